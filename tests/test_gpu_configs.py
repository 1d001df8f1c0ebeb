"""GPU parity at the BASELINE.json configurations' real WIDTHS and token schedules (depth reduced so that the CPU
oracle finishes in seconds), plus size-independent properties at full size:

  cfg2  8 frames, token_kept_ratio 0.3 -> r = 15, 264 tokens/frame, prefix 2142
  cfg3 16 frames, ratio 0.2            -> r = 18, 171 tokens/frame (merge-dominated)
  cfg5  8 frames, ratio 0.8            -> r = 4, 605 tokens/frame, 2048 new tokens (long KV, many pages / splits)
"""
import numpy as np
import pytest
import torch

from oracle import aurora_oracle as O
from oracle import tome_ref
from tests.parity_bounds import FEAT_TOL, FREE_RUN_AGREE, FREE_RUN_DRIFT, LOGIT_TOL_DEEP, LOGIT_TOL_WIDE, NEAR_TIE
from tests.util import observe, rand_llm_weights, rand_vit_weights, rel_l2, to_match

pytestmark = pytest.mark.gpu

VIT_H = dict(hidden_size=1280, num_attention_heads=16, num_hidden_layers=3, intermediate_size=5120, patch_size=14,
             image_size=378, hidden_act="quick_gelu", layer_norm_eps=1e-5)
LLAMA_7B_WIDTH = dict(hidden_size=4096, num_attention_heads=32, num_hidden_layers=2, intermediate_size=11008, vocab_size=32000,
                      rms_norm_eps=1e-5, rope_theta=1e4, rope_factor=4.0)


def test_schedule_of_every_baseline_config():
    from aurora_amd.engine import tokens_at_layer, tome_r
    for ratio, r, kept in ((0.3, 15, 264), (0.2, 18, 171), (0.8, 4, 605), (1.0, 0, 729)):
        assert tome_r(378, 378, 14, ratio, 32) == r
        assert tokens_at_layer(730, r, 31) - 1 == kept


@pytest.mark.parametrize("ratio,frames", [(0.3, 2), (0.2, 2), (0.8, 1)])
def test_vit_h_width_layers_teacher_forced(ratio, frames):
    """ViT-H/14-378 width (D 1280, 16 x 80, MLP 5120, 730 tokens): two real layers per config ratio, teacher-forced."""
    from aurora_amd.engine import AuroraCapEngine, tome_r
    w = rand_vit_weights(VIT_H, 7, wstd=0.02)
    eng = AuroraCapEngine({"vit": VIT_H, "llm": None}, {"vit": w}, max_frames=frames, max_batch=1, max_ctx=128, max_new_tokens=8)
    try:
        r = tome_r(378, 378, 14, ratio, 32)
        x = torch.randn(frames, 730, 1280, generator=torch.Generator().manual_seed(5)).half().float()
        size = None
        for layer in range(2):
            xo, so, metric, idx = eng.vit_layer(layer, x, size, r)
            mc = tome_ref.match(metric.cpu().numpy(), r)
            for k in ("node_idx", "unm_idx", "src_idx", "dst_idx"):
                np.testing.assert_array_equal(idx[k].cpu().numpy(), mc[k], err_msg=f"layer {layer} {k}")
            xr, sr = O.vit_layer(x, size, w["layers"][layer], 16, r, "quick_gelu", forced_match=to_match(idx))
            observe("configs/vit_h_single_layer_rel_l2", rel_l2(xo.float().cpu(), xr), FEAT_TOL)
            assert xo.shape[1] == x.shape[1] - r
            x, size = xo.float().cpu(), so.cpu()[..., None]
    finally:
        eng.close()


def test_llama_7b_width_prefill_and_decode_logits():
    """Llama-7B width (d 4096, 32 x 128, MLP 11008, vocab 32000), 2 layers: prefill of a 2142-row prefix shape is too slow
    for the CPU oracle, so a 300-token prefix checks logits; the 2142-token prefix checks bitwise properties."""
    from aurora_amd.engine import AuroraCapEngine
    from tests.test_gpu_llm import LOGIT_TOL, padded, teacher_forced_logits
    cfg = LLAMA_7B_WIDTH
    w = rand_llm_weights(cfg, 3, wstd=0.02)
    eng = AuroraCapEngine({"vit": None, "llm": cfg}, {"llm": w}, max_frames=1, max_batch=4, max_ctx=2560, max_new_tokens=16)
    try:
        emb = (torch.randn(300, 4096, generator=torch.Generator().manual_seed(1)) * 0.5).half().float()
        eng.begin_batch(1, 6, None)
        eng.prefill(0, padded(emb), 300)
        logits = [eng.logits()[0].cpu()]
        for _ in range(5):
            eng.decode(1)
            logits.append(eng.logits()[0].cpu())
        ids = eng.outputs()[0]
        ref = teacher_forced_logits(emb, ids, w, cfg)
        scale = ref.abs().max().item()
        for i in range(6):
            observe("configs/llama7b_width_2_layers_logits_over_scale", (logits[i] - ref[i]).abs().max().item() / scale, LOGIT_TOL_WIDE)
        # cfg2 prefix length (2142): batched prefill == per-slot prefill, graph decode deterministic
        g = torch.Generator().manual_seed(2)
        embs = [(torch.randn(2142, 4096, generator=g) * 0.5).half() for _ in range(3)]
        big = torch.cat([padded(e.float()) for e in embs], 0).contiguous()
        eng.begin_batch(3, 16, None)
        eng.prefill_batch(0, 3, big, 2142)
        first = eng.logits()[:3].cpu()                              # logits of the LAST prompt position of every sequence
        eng.decode(15)
        a = eng.outputs()
        # the 2142-row prefill (cfg2's prefix length: 67 query blocks, 34 pages) had only ever met bitwise
        # properties.  One sequence through the fp32 oracle at this width: its last-position logits are the first token's
        h, _ = O.llama_forward(embs[0].float(), w, cfg, None, 0)
        ref_first = torch.nn.functional.linear(h[-1:], w["lm_head.weight"])[0]
        observe("configs/llama7b_width_2142_row_prefill_first_token_logits_over_scale",
                (first[0] - ref_first).abs().max().item() / ref_first.abs().max().item(), LOGIT_TOL_WIDE)
        if (ref_first.topk(2).values[0] - ref_first.topk(2).values[1]).item() > 2 * LOGIT_TOL_WIDE * ref_first.abs().max().item():
            assert a[0][0] == int(ref_first.argmax())
        eng.begin_batch(3, 16, None)
        for b in range(3):
            eng.prefill(b, padded(embs[b].float()), 2142)
        eng.decode(15)
        assert eng.outputs() == a and all(len(x) == 16 for x in a)
    finally:
        eng.close()


@pytest.mark.parametrize("name,L", [("cfg3", 2766), ("cfg5", 4870)])
def test_llama_7b_width_prefill_at_the_other_configs_prefix_lengths_vs_fp32(name, L):
    """cfg3 / cfg5 at full size had met property checks only.  Their prefix lengths (16 x 171 + 30 = 2766 and
    8 x 605 + 30 = 4870 rows: 44 / 77 pages, 22 / 39 causal query blocks) through 2 layers at Llama-7B width: the first token's logits
    (last prompt position) and the five decode steps after it against one teacher-forced fp32 pass of the oracle."""
    from aurora_amd.engine import AuroraCapEngine
    from tests.test_gpu_llm import padded, teacher_forced_logits
    cfg = LLAMA_7B_WIDTH
    w = rand_llm_weights(cfg, 3, wstd=0.02)
    eng = AuroraCapEngine({"vit": None, "llm": cfg}, {"llm": w}, max_frames=1, max_batch=1, max_ctx=(L + 16 + 63) // 64 * 64, max_new_tokens=8)
    try:
        emb = (torch.randn(L, 4096, generator=torch.Generator().manual_seed(len(name) + L)) * 0.5).half().float()
        eng.begin_batch(1, 6, None)
        eng.prefill(0, padded(emb), L)
        logits = [eng.logits()[0].cpu()]
        for _ in range(5):
            eng.decode(1)
            logits.append(eng.logits()[0].cpu())
        ids = eng.outputs()[0]
    finally:
        eng.close()
    ref = teacher_forced_logits(emb, ids, w, cfg)
    scale = ref.abs().max().item()
    for i in range(6):
        observe(f"configs/llama7b_width_{name}_prefix_logits_over_scale", (logits[i] - ref[i]).abs().max().item() / scale, LOGIT_TOL_WIDE)


def test_long_kv_decode_many_pages_cfg5_shape():
    """cfg5 shape: prefix 4870, 2048 new tokens (context grows to 6.9k = 108 pages, 27 attention splits) on a small-width
    model: teacher-forced logits at the start, middle and end of the generation, graph == eager."""
    from aurora_amd.engine import AuroraCapEngine
    from tests.test_gpu_llm import LLM_CFGS, LOGIT_TOL, padded
    cfg = LLM_CFGS["hd32"]
    w = rand_llm_weights(cfg, 8)
    L0, N = 4870, 2048
    emb = torch.randn(L0, cfg["hidden_size"], generator=torch.Generator().manual_seed(6)).half().float()
    outs = []
    for use_graph in (True, False):
        eng = AuroraCapEngine({"vit": None, "llm": cfg}, {"llm": w}, max_frames=1, max_batch=1, max_ctx=7040, max_new_tokens=N,
                              use_graph=use_graph)
        try:
            eng.begin_batch(1, N, None)
            eng.prefill(0, padded(emb), L0)
            if use_graph:
                eng.decode(N - 2)
                eng.decode(1)                      # last step separately: logits of the final position
                last_logits = eng.logits()[0].cpu()
            else:
                eng.decode(N - 1)
            outs.append(eng.outputs()[0])
        finally:
            eng.close()
    assert len(outs[0]) == N and outs[0] == outs[1]
    ids = outs[0]
    full = torch.cat([emb, w["embed_tokens.weight"][torch.tensor(ids[:-1], dtype=torch.long)]], 0)
    h, _ = O.llama_forward(full, w, cfg, None, 0)
    ref_last = torch.nn.functional.linear(h[-1:], w["lm_head.weight"])[0]
    observe("configs/cfg5_shape_last_logits_over_scale", (last_logits - ref_last).abs().max().item() / ref_last.abs().max().item(), LOGIT_TOL)
    # every generated token is the first argmax of logits that agree with the oracle wherever its margin is clear
    ref_all = torch.nn.functional.linear(h[L0 - 1:], w["lm_head.weight"])
    top2 = ref_all.topk(2, dim=-1).values
    clear = (top2[:, 0] - top2[:, 1]) > 2 * LOGIT_TOL * ref_all.abs().max()
    agree = (ref_all.argmax(-1) == torch.tensor(ids))
    assert bool(agree[clear].all()) and int(clear.sum()) > N // 2


def test_full_depth_cfg2_and_cfg3_batch_invariance_and_repeatability():
    """FULL-size AuroraCap-7B (ViT-H/14-378 x 31 layers, Llama-7B x 32 layers, seeded synthetic weights): the CPU oracle
    cannot run this in test time, so the size-independent properties are checked instead - captioning three clips in one
    batch (two share a prefill pass, one has 16 frames at ratio-0.2-like length) gives every clip exactly the ids it
    gets alone, twice in a row, and every id is a valid vocabulary index."""
    from aurora_amd import synthetic as S
    from aurora_amd.engine import AuroraCapEngine
    cfg = S.AURORACAP_7B
    w = {"vit": S.vit_weights(cfg["vit"]), "projector": S.projector_weights(1280, 4096), "llm": S.llm_weights(cfg["llm"])}
    eng = AuroraCapEngine(cfg, w, max_frames=16, max_batch=3, max_ctx=4608, max_new_tokens=12)
    del w
    torch.cuda.empty_cache()
    try:
        clips = [(S.frames(8, 0), S.prompt_ids(8, 0)), (S.frames(8, 1), S.prompt_ids(8, 1)), (S.frames(16, 2), S.prompt_ids(16, 2))]
        together = eng.caption_batch(clips, 0.3, 12, eos_id=None)
        again = eng.caption_batch(clips, 0.3, 12, eos_id=None)
        assert together == again
        for b, (px, ids) in enumerate(clips):
            alone = eng.caption_ids(px, ids, 0.3, 12, eos_id=None)
            assert together[b] == alone, b
            assert len(alone) == 12 and all(0 <= t < 32000 for t in alone)
        assert together[0] != together[1]                                       # different clips, different captions
        # continuous batching at full size: 3 clips through 2 KV slots, EOS = a token that occurs in clip 1's caption
        eos = together[1][4]
        want = [t[: t.index(eos) + 1] if eos in t else t for t in together]
        got = dict(eng.caption_stream(clips, 0.3, 12, eos_id=eos, slots=2, check_every=4))
        assert [got[i] for i in range(3)] == want
    finally:
        eng.close()


@pytest.mark.parametrize("norm_std,bias_std", [(0.0, 0.0), (0.1, 0.05)])
def test_vit_h_full_depth_every_layer_teacher_forced(norm_std, bias_std):
    """The complete vision tower at its real size (ViT-H/14-378, all 31 evaluated layers, r = 15: 730 -> 265 tokens) for one
    frame.  Run end to end, fp16 and fp32 arithmetic pick different merge pairs at some near tie within 31 steps and the
    token sets drift apart (measured: 6.5 % on the mean feature), so every layer is checked on ITS OWN input
    instead: indices bit-exact against the C oracle fed with the GPU's metric, layer output against the CPU oracle with
    the same merge forced, then the GPU output becomes the next layer's input."""
    from aurora_amd import synthetic as S
    from aurora_amd.engine import AuroraCapEngine
    v = S.AURORACAP_7B["vit"]
    wg = S.vit_weights(v, norm_std=norm_std, bias_std=bias_std)      # (0.1, 0.05): LayerNorm weights / every bias away from 1 / 0
    f32 = lambda d: {k: ([f32(x) for x in val] if isinstance(val, list) else val.float().cpu()) for k, val in d.items()}
    w = f32(wg)
    eng = AuroraCapEngine({"vit": v, "llm": None}, {"vit": wg}, max_frames=1, max_batch=1, max_ctx=128, max_new_tokens=8)
    try:
        r = eng.tome_r(0.3)
        assert r == 15
        x = O.vit_embed(S.frames(1, 3).float().cpu(), w, 14, v.get("layer_norm_eps", 1e-5)).half().float()
        size, worst = None, 0.0
        for layer in range(v["num_hidden_layers"] - 1):
            xo, so, metric, idx = eng.vit_layer(layer, x, size, r)
            mc = tome_ref.match(metric.cpu().numpy(), r)
            for k in ("node_idx", "unm_idx", "src_idx", "dst_idx"):
                np.testing.assert_array_equal(idx[k].cpu().numpy(), mc[k], err_msg=f"layer {layer} {k}")
            xr, _ = O.vit_layer(x, size, w["layers"][layer], v["num_attention_heads"], r, v["hidden_act"], forced_match=to_match(idx))
            err = rel_l2(xo.float().cpu(), xr)
            worst = max(worst, err)
            observe("configs/vit_h_every_layer_teacher_forced_rel_l2", err, FEAT_TOL)
            assert xo.shape[1] == x.shape[1] - r
            x, size = xo.float().cpu(), so.cpu()[..., None]
        assert x.shape[1] == 265                                                  # 264 patch tokens + CLS enter layer 31
    finally:
        eng.close()


@pytest.mark.parametrize("norm_std", [0.0, 0.1])
def test_llama_7b_full_depth_teacher_forced_logits(norm_std):
    """Vicuna/Llama-7B at its real size (32 layers x d 4096 x MLP 11008, vocab 32000, linear RoPE scaling 4) with the seeded
    synthetic weights: a 96-token prefix + 5 greedy tokens; the CPU oracle replays the GPU's own tokens in one fp32 pass
    (teacher forcing) and its logits must match at every generated position within the tolerance of the small-model
    tests - the norm-free decode step and the paged KV path are exercised at full depth."""
    from aurora_amd import synthetic as S
    from aurora_amd.engine import AuroraCapEngine
    from tests.test_gpu_llm import LOGIT_TOL, padded, teacher_forced_logits
    cfg = S.VICUNA_7B_16K
    # norm_std 0.1: RMSNorm weights 1 + N(0, 0.1^2) - the FOLDED-norm decode path (W' = W diag(w_norm), rounded to fp16 at pack
    # time) at full depth with weights as far from 1 as a real checkpoint's
    wg = S.llm_weights(cfg, norm_std=norm_std)
    eng = AuroraCapEngine({"vit": None, "llm": cfg}, {"llm": wg}, max_frames=1, max_batch=1, max_ctx=256, max_new_tokens=8)
    try:
        emb = (torch.randn(96, 4096, generator=torch.Generator().manual_seed(4)) * 0.02).half().float()
        eng.begin_batch(1, 6, None)
        eng.prefill(0, padded(emb), 96)
        logits = [eng.logits()[0].cpu()]
        for _ in range(5):
            eng.decode(1)
            logits.append(eng.logits()[0].cpu())
        ids = eng.outputs()[0]
    finally:
        eng.close()
    f32 = lambda d: {k: ([f32(x) for x in val] if isinstance(val, list) else val.float().cpu()) for k, val in d.items()}
    w = f32(wg)
    del wg
    ref = teacher_forced_logits(emb, ids, w, cfg)
    scale = ref.abs().max().item()
    for i in range(6):
        observe("configs/llama7b_full_depth_logits_over_scale", (logits[i] - ref[i]).abs().max().item() / scale, LOGIT_TOL_DEEP)


def test_projector_and_splice_at_real_size():
    """Projector 1280 -> 4096 -> 4096 (erf GELU) + prefix splice for cfg2's 8 x 264 visual tokens and a 30-token prompt."""
    from aurora_amd import synthetic as S
    from aurora_amd.engine import AuroraCapEngine
    cfg = dict(LLAMA_7B_WIDTH, num_hidden_layers=1)
    w = {"projector": {k: v.float().cpu() for k, v in S.projector_weights(1280, 4096).items()}, "llm": rand_llm_weights(cfg, 3, wstd=0.02)}
    eng = AuroraCapEngine({"vit": VIT_H, "llm": cfg}, w, max_frames=8, max_batch=1, max_ctx=2304, max_new_tokens=8)   # vit: dims only
    try:
        vis = (torch.randn(8, 264, 1280, generator=torch.Generator().manual_seed(2)) * 0.5).half()
        ids = S.prompt_ids(8, 0)
        emb, L = eng.project_splice(vis, ids)
        ref = O.splice(torch.tensor(ids), w["llm"]["embed_tokens.weight"],
                       O.projector(vis.float().reshape(1, -1, 1280), w["projector"]).reshape(8, 264, -1))
        assert L == ref.shape[0] == 30 + 8 * 264
        observe("configs/project_splice_real_size_rel_l2", rel_l2(emb[:L].float().cpu(), ref), FEAT_TOL)
        text_rows = [i for i, t in enumerate(ids) if t != -200]
        row = 0
        for t in ids:                                               # text rows are exact copies of the embedding table
            if t == -200:
                row += 264
            else:
                assert torch.equal(emb[row].cpu(), w["llm"]["embed_tokens.weight"][t].half())
                row += 1
    finally:
        eng.close()


def test_vit_h_free_running_index_audit():
    """SURVEY 8c parity contract (iii), at the real size: the whole vision tower FREE-RUNNING on the GPU (each layer consumes the
    previous layer's GPU output, nothing is teacher-forced) against the oracle's own free-running chain with fp16 storage
    (ViT-H/14-378, 31 evaluated layers, r = 15, two frames).  Per frame, layer by layer while the two chains still hold the same
    token set: the merge indices are compared; the first disagreement of a frame is AUDITED - the pair that differs must be a
    near tie of the oracle's fp32 scores on the oracle's own state (a different source token within 2e-3 of the r-th best
    node_max, or a different destination within 2e-3 of the best score of that row) - after which the frame's token sets differ by
    construction and it leaves the comparison.  Prints the frame-layer agreement rate and every audited flip; asserts that no
    flip is anything but a near tie and that the final features still agree to the documented drift."""
    from aurora_amd import synthetic as S
    from aurora_amd.engine import AuroraCapEngine
    v = S.AURORACAP_7B["vit"]
    wg = S.vit_weights(v, norm_std=0.1, bias_std=0.05)
    f32 = lambda d: {k: ([f32(x) for x in val] if isinstance(val, list) else val.float().cpu()) for k, val in d.items()}
    w = f32(wg)
    F = 2
    px = S.frames(F, 11).float().cpu()
    cap = []
    feats_ref = O.vit_features(px, w, v, 0.3, O.fp16_storage, capture=cap)                 # the oracle's free run
    eng = AuroraCapEngine({"vit": v, "llm": None}, {"vit": wg}, max_frames=F, max_batch=1, max_ctx=128, max_new_tokens=8)
    try:
        r = eng.tome_r(0.3)
        x = O.vit_embed(px, w, 14, v.get("layer_norm_eps", 1e-5)).half().float()
        size = None
        alive = [True] * F
        agree = compared = 0
        flips = []
        L = v["num_hidden_layers"] - 1
        for layer in range(L):
            xo, so, metric, idx = eng.vit_layer(layer, x, size, r)
            mo = cap[layer]["match"]
            m = cap[layer]["metric"].float()
            for f in range(F):
                if not alive[f]:
                    continue
                compared += 1
                g_src, g_dst = idx["src_idx"][f].cpu().tolist(), idx["dst_idx"][f].cpu().tolist()
                o_src, o_dst = mo["src_idx"][f].tolist(), mo["dst_idx"][f].tolist()
                if g_src == o_src and g_dst == o_dst and idx["unm_idx"][f].cpu().tolist() == mo["unm_idx"][f].tolist():
                    agree += 1
                    continue
                alive[f] = False                                                            # first flip of this frame: audit it
                mh = m[f] / m[f].norm(dim=-1, keepdim=True)
                sc = mh[0::2] @ mh[1::2].T
                sc[0] = -np.inf
                nmax = sc.max(-1).values
                order = torch.argsort(nmax, descending=True, stable=True)
                boundary = 0.5 * (nmax[order[r - 1]].item() + nmax[order[r]].item())
                top2 = sc.topk(2, dim=-1).values
                worst = 0.0
                for a in set(g_src) ^ set(o_src):                                           # a different source token: at the r boundary
                    worst = max(worst, abs(nmax[a].item() - boundary))
                gd = dict(zip(g_src, g_dst))
                for a, d_o in zip(o_src, o_dst):                                            # same source, different partner: a tie of its row
                    if a in gd and gd[a] != d_o:
                        worst = max(worst, (top2[a, 0] - top2[a, 1]).item())
                if set(g_src) == set(o_src) and all(gd[a] == d for a, d in zip(o_src, o_dst)):
                    # same pairs, different rank ORDER among the sources: node_max values that tie in rank
                    pos_o = {a: i for i, a in enumerate(o_src)}
                    for i, a in enumerate(g_src):
                        if pos_o[a] != i:
                            worst = max(worst, abs(nmax[a].item() - nmax[o_src[i]].item()))
                flips.append((layer, f, worst))
                observe("configs/vit_h_free_run_flip_score_gap", worst, NEAR_TIE)        # an index difference must be a near tie of the fp32 scores
            x, size = xo.float().cpu(), so.cpu()[..., None]
        rate = agree / max(compared, 1)
        feats = x[:, 1:]
        drift = rel_l2(feats, feats_ref)
        mean_drift = rel_l2(feats.mean(1), feats_ref.mean(1))
        print(f"\nViT-H free run: indices agree on {agree} of {compared} comparable frame-layers ({100 * rate:.1f} %); first flips (layer, frame, "
              f"score gap): {[(l, f, round(g, 6)) for l, f, g in flips]}; frames still identical at layer {L}: {sum(alive)} of {F}; "
              f"final feature rel-L2 {drift:.4f}, mean-feature rel-L2 {mean_drift:.4f}")
        assert compared >= F
        observe("configs/vit_h_free_run_agreeing_frame_layers", agree, FREE_RUN_AGREE, at_least=True)
        assert feats.shape == feats_ref.shape
        observe("configs/vit_h_free_run_mean_feature_drift", mean_drift, FREE_RUN_DRIFT)   # once a near tie has flipped the token sets differ; identical chains give < FEAT_TOL
        if all(alive):
            observe("configs/vit_h_free_run_identical_chain_rel_l2", drift, FEAT_TOL)
    finally:
        eng.close()


@pytest.mark.parametrize("name,frames,ratio,kept", [("cfg3", 16, 0.2, 171), ("cfg5", 8, 0.8, 605)])
def test_full_size_cfg3_and_cfg5_at_their_own_ratios(name, frames, ratio, kept):
    """BASELINE configs[2] (16 frames, ratio 0.2 -> r = 18, 171 tokens per frame) and configs[4] (8 frames, ratio 0.8 -> r = 4, 605
    tokens per frame, prefix 4870) at FULL size and at THEIR OWN merge ratios (the full-size property test used ratio
    0.3 for its 16-frame clip): token schedule as the reference computes it, a batch of two clips == each clip alone == a repeat,
    the overlapped continuous-batching stream == the same, 8 greedy tokens, ids inside the vocabulary."""
    from aurora_amd import synthetic as S
    from aurora_amd.engine import AuroraCapEngine, tokens_at_layer
    cfg = S.AURORACAP_7B
    w = {"vit": S.vit_weights(cfg["vit"]), "projector": S.projector_weights(1280, 4096), "llm": S.llm_weights(cfg["llm"])}
    L0 = 30 + frames * kept
    eng = AuroraCapEngine(cfg, w, max_frames=2 * frames, max_batch=2, max_ctx=-(-(L0 + 8) // 64) * 64, max_new_tokens=8, spare_slots=1)
    del w
    torch.cuda.empty_cache()
    try:
        r = eng.tome_r(ratio)
        assert tokens_at_layer(730, r, 31) - 1 == kept
        clips = [(S.frames(frames, 20 + i), S.prompt_ids(frames, 20 + i)) for i in range(2)]
        vis = eng.vit_encode(clips[0][0], r)
        assert tuple(vis.shape) == (frames, kept, 1280)
        alone = [eng.caption_ids(px, ids, ratio, 8, eos_id=None) for px, ids in clips]
        assert all(len(a) == 8 and all(0 <= t < 32000 for t in a) for a in alone) and alone[0] != alone[1]
        assert eng.caption_batch(clips, ratio, 8, eos_id=None) == alone
        assert eng.caption_batch(clips, ratio, 8, eos_id=None) == alone
        got = dict(eng.caption_stream(clips + clips[:1], ratio, 8, eos_id=None, check_every=4))      # front ends prefetched beside the decode
        assert [got[i] for i in range(3)] == alone + alone[:1]
    finally:
        eng.close()


def test_full_size_configs3_per_gpu_share_eight_clips_in_one_batch():
    """BASELINE configs[3]'s per-GPU workload (`bench.py --gpus 8 --config cfg4`): EIGHT cfg2 clips (8 frames, ratio 0.3, prefix 2142) in
    ONE batch of an 8-slot engine at full model size, 256 new tokens - the regime where the decode step is the weight stream, not K / V
    (decode attention runs 2 splits per (sequence, head); the projections take the per-wave-x structure of engines <= 32 slots).
    Properties: the batch == each of its clips alone (two of them are re-run alone in full), a repeat of the batch is bit-identical,
    every clip yields exactly 256 ids inside the vocabulary and the eight captions are distinct; clip i of rank k of the 8-rank job is
    clip k + 8 i (round-robin shard) - the batch below is rank 3's."""
    from aurora_amd import parallel
    from aurora_amd import synthetic as S
    from aurora_amd.engine import AuroraCapEngine
    cfg = S.AURORACAP_7B
    w = {"vit": S.vit_weights(cfg["vit"]), "projector": S.projector_weights(1280, 4096), "llm": S.llm_weights(cfg["llm"])}
    N, L0 = 256, 30 + 8 * 264
    eng = AuroraCapEngine(cfg, w, max_frames=8 * 8, max_batch=8, max_ctx=-(-(L0 + N) // 64) * 64, max_new_tokens=N)
    del w
    torch.cuda.empty_cache()
    try:
        mine = parallel.shard_clips(64, 3, 8)
        assert mine == [3, 11, 19, 27, 35, 43, 51, 59]
        clips = [(S.frames(8, c), S.prompt_ids(8, c)) for c in mine]
        batch = eng.caption_batch(clips, 0.3, N, eos_id=None)
        assert len(batch) == 8 and all(len(o) == N and all(0 <= t < 32000 for t in o) for o in batch)
        assert len({tuple(o) for o in batch}) == 8
        assert eng.caption_batch(clips, 0.3, N, eos_id=None) == batch                 # repeatable, graph replays included
        for j in (0, 5):                                                               # batch == alone, all 256 tokens
            assert eng.caption_ids(clips[j][0], clips[j][1], 0.3, N, eos_id=None) == batch[j]
    finally:
        eng.close()


@pytest.mark.parametrize("name,frames,ratio,kept,new", [("cfg3", 16, 0.2, 171, 512), ("cfg5", 8, 0.8, 605, 2048)])
def test_full_size_cfg3_and_cfg5_at_their_full_generation_lengths(name, frames, ratio, kept, new):
    """BASELINE configs[2] / configs[4] at full model size AND their full generation lengths (512 / 2048 new tokens; VERDICT r3: the
    full-size property test stopped at 8 tokens, the full lengths ran at small width only): two clips in a 2-slot engine, hipGraph
    decode over a context that grows to 3278 / 6918 tokens (52 / 109 KV pages per sequence).  Properties: every clip yields exactly
    `new` ids inside the vocabulary, the batch == clip 0 alone over ALL positions (batch-invariant kernels, page-table walk included),
    a repeat is bit-identical, and the two captions differ."""
    from aurora_amd import synthetic as S
    from aurora_amd.engine import AuroraCapEngine
    cfg = S.AURORACAP_7B
    w = {"vit": S.vit_weights(cfg["vit"]), "projector": S.projector_weights(1280, 4096), "llm": S.llm_weights(cfg["llm"])}
    L0 = 30 + frames * kept
    eng = AuroraCapEngine(cfg, w, max_frames=2 * frames, max_batch=2, max_ctx=-(-(L0 + new) // 64) * 64, max_new_tokens=new)
    del w
    torch.cuda.empty_cache()
    try:
        clips = [(S.frames(frames, 40 + i), S.prompt_ids(frames, 40 + i)) for i in range(2)]
        batch = eng.caption_batch(clips, ratio, new, eos_id=None)
        assert [len(o) for o in batch] == [new, new] and all(0 <= t < 32000 for o in batch for t in o) and batch[0] != batch[1]
        assert eng.caption_ids(clips[0][0], clips[0][1], ratio, new, eos_id=None) == batch[0]
        assert eng.caption_batch(clips, ratio, new, eos_id=None) == batch
    finally:
        eng.close()
