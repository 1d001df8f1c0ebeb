"""GPU parity: ViT encoder layers with per-layer ToMe, through the C ABI, vs the CPU oracle.

Contract: (i) indices bit-exact vs the C oracle on the GPU's own metric bytes; (iii) features
rel-L2 <= 5e-3 vs the fp32 oracle when the merges agree - so every layer is teacher-forced (oracle and
GPU get the same fp16 input; the oracle is told the GPU's match) and the index agreement between the
fp32 oracle's own choice and the GPU's is reported / bounded separately."""
import numpy as np
import pytest
import torch

from oracle import aurora_oracle as O
from oracle import tome_ref
from tests.parity_bounds import FEAT_TOL
from tests.util import observe, rand_vit_weights, rel_l2, to_match

pytestmark = pytest.mark.gpu

CFGS = {
    # name: (cfg, frames, r)
    "tiny_hd16": (dict(hidden_size=64, num_attention_heads=4, num_hidden_layers=4, intermediate_size=128, patch_size=14,
                       image_size=56, hidden_act="quick_gelu", layer_norm_eps=1e-5), 2, 2),
    "hd80_gelu": (dict(hidden_size=320, num_attention_heads=4, num_hidden_layers=3, intermediate_size=640, patch_size=14,
                       image_size=112, hidden_act="gelu", layer_norm_eps=1e-5), 3, 5),
    "mid_t730": (dict(hidden_size=320, num_attention_heads=4, num_hidden_layers=4, intermediate_size=640, patch_size=14,
                      image_size=378, hidden_act="quick_gelu", layer_norm_eps=1e-5), 2, 15),
}


def make_engine(cfg, frames, seed):
    from aurora_amd.engine import AuroraCapEngine
    w = rand_vit_weights(cfg, seed)
    e = AuroraCapEngine({"vit": cfg, "llm": None}, {"vit": w}, max_frames=frames, max_batch=1, max_ctx=128, max_new_tokens=8)
    return e, w


@pytest.mark.parametrize("name", list(CFGS))
def test_vit_layer_teacher_forced(name):
    cfg, frames, r = CFGS[name]
    eng, w = make_engine(cfg, frames, 42)
    try:
        heads = cfg["num_attention_heads"]
        t = (cfg["image_size"] // cfg["patch_size"]) ** 2 + 1
        gen = torch.Generator().manual_seed(7)
        x = torch.randn(frames, t, cfg["hidden_size"], generator=gen).half().float()
        size = None
        agree = []
        for layer in range(cfg["num_hidden_layers"] - 1):
            xo, so, metric, idx = eng.vit_layer(layer, x, size, r)
            # (i) bit-exact indices on the GPU's own metric bytes
            mc = tome_ref.match(metric.cpu().numpy(), r)
            for k in ("node_idx", "unm_idx", "src_idx", "dst_idx"):
                np.testing.assert_array_equal(idx[k].cpu().numpy(), mc[k], err_msg=f"layer {layer} {k}")
            # metric itself vs fp32 oracle (fp16 K storage): tolerance
            cap = []
            xr, sr = O.vit_layer(x, size, w["layers"][layer], heads, r, cfg["hidden_act"], forced_match=to_match(idx), capture=cap)
            np.testing.assert_allclose(metric.cpu().numpy(), cap[0]["metric"].numpy(), rtol=5e-3, atol=5e-3)
            # (iii) features with forced (= GPU) merges
            observe(f"vit/layer_teacher_forced_rel_l2[{name}]", rel_l2(xo.float().cpu(), xr), FEAT_TOL)
            np.testing.assert_array_equal(so.cpu().numpy(), sr[..., 0].numpy())
            # agreement of the fp32 oracle's own src set with the GPU's (near-tie audit)
            own = O.bipartite_match(cap[0]["metric"], r)
            for f in range(frames):
                a, b = set(own["src_idx"][f].tolist()), set(idx["src_idx"][f].cpu().tolist())
                agree.append(len(a & b) / max(1, len(a)))
            x, size = xo.float().cpu(), so.cpu()[..., None]
        assert np.mean(agree) > 0.9, agree
    finally:
        eng.close()


@pytest.mark.parametrize("name", ["tiny_hd16", "hd80_gelu"])
def test_vit_encode_end_to_end(name):
    """Whole tower: pixels -> hidden_states[-2][:, 1:].  Compared with the oracle run with fp16 storage
    emulation; if a near-tie makes the two pick different merges the row sets differ, so the bound is on
    the per-frame mean feature (permutation-insensitive) plus exact equality of the token count."""
    cfg, frames, r = CFGS[name]
    eng, w = make_engine(cfg, frames, 43)
    try:
        gen = torch.Generator().manual_seed(9)
        px = torch.randn(frames, 3, cfg["image_size"], cfg["image_size"], generator=gen).half().float()
        ratio = 0.5
        rr = eng.tome_r(ratio)
        assert rr == O.tome_r(cfg["image_size"], cfg["image_size"], cfg["patch_size"], ratio, cfg["num_hidden_layers"])
        out = eng.vit_encode(px, rr).float().cpu()
        ref = O.vit_features(px, w, cfg, ratio, q=O.fp16_storage)
        assert out.shape == ref.shape
        observe(f"vit/encode_end_to_end_mean_feature_rel_l2[{name}]", rel_l2(out.mean(1), ref.mean(1)), 1e-2)
        if rel_l2(out, ref) > 1e-2:          # merges diverged at a near-tie: rows are permuted, compare sorted norms
            assert rel_l2(out.norm(dim=-1).sort(-1).values, ref.norm(dim=-1).sort(-1).values) < 2e-2
    finally:
        eng.close()


def test_vit_encode_ratio_one_exact_schedule():
    """token_kept_ratio = 1.0 -> r = 0: no merging, all 16 patches per frame, plain ViT numerics."""
    cfg, frames, _ = CFGS["tiny_hd16"]
    eng, w = make_engine(cfg, frames, 44)
    try:
        gen = torch.Generator().manual_seed(10)
        px = torch.randn(frames, 3, 56, 56, generator=gen).half().float()
        out = eng.vit_encode(px, eng.tome_r(1.0)).float().cpu()
        ref = O.vit_features(px, w, cfg, 1.0)
        assert out.shape == (frames, 16, 64)
        observe("vit/encode_ratio_one_rel_l2", rel_l2(out, ref), FEAT_TOL)
    finally:
        eng.close()


def test_vit_encode_double_run_bitwise():
    cfg, frames, r = CFGS["hd80_gelu"]
    eng, _ = make_engine(cfg, frames, 45)
    try:
        px = torch.randn(frames, 3, 112, 112, generator=torch.Generator().manual_seed(1)).half()
        a = eng.vit_encode(px, r).clone()
        b = eng.vit_encode(px, r)
        assert torch.equal(a, b)
    finally:
        eng.close()


@pytest.mark.parametrize("name", ["tiny_hd16", "hd80_gelu", "mid_t730"])
def test_vit_layer_gemm256_equals_gemm128(name):
    """The 256x256 kernel's QKV-fragment / V^T epilogues give the same layer output as the 128x128 kernel's."""
    cfg, frames, r = CFGS[name]
    eng, w = make_engine(cfg, frames, 46)
    try:
        t = (cfg["image_size"] // cfg["patch_size"]) ** 2 + 1
        x = torch.randn(frames, t, cfg["hidden_size"], generator=torch.Generator().manual_seed(3)).half()
        outs = []
        for mode in (0, 2):
            eng.set_option("gemm_mode", mode)
            xo, so, metric, idx = eng.vit_layer(0, x, None, r)
            outs.append((xo.clone(), metric.clone(), idx["src_idx"].clone()))
        eng.set_option("gemm_mode", 1)
        assert torch.equal(outs[0][1], outs[1][1])          # K fragments -> metric
        assert torch.equal(outs[0][2], outs[1][2])
        assert torch.equal(outs[0][0], outs[1][0])
    finally:
        eng.set_option("gemm_mode", 1)
        eng.close()


@pytest.mark.parametrize("name", ["tiny_hd16", "hd80_gelu", "mid_t730"])
def test_layernorm2_out_of_the_merge_launch_is_bitwise_the_separate_launch(name):
    """Round 5: a merging layer's LayerNorm 2 (aurora.py:750) is written by the ToMe select + merge launch from the registers that hold
    the merged row (tome.hip), instead of a norm_kernel pass that reads it back.  Same operations in the same order on the rounded fp16
    row: every layer output (and with it the size vector and the indices) must be BITWISE that of the two-launch form
    (`tome_fused_ln` 0), with and without a carried size vector, pad rows included (they feed the next GEMM's tiles)."""
    cfg, frames, r = CFGS[name]
    eng, _ = make_engine(cfg, frames, 11)
    try:
        t = (cfg["image_size"] // cfg["patch_size"]) ** 2 + 1
        gen = torch.Generator().manual_seed(3)
        x = torch.randn(frames, t, cfg["hidden_size"], generator=gen).half().float()
        size = None
        for layer in range(cfg["num_hidden_layers"] - 1):
            outs = []
            for fused in (1, 0, 1):
                eng.set_option("tome_fused_ln", fused)
                xo, so, metric, idx = eng.vit_layer(layer, x, size, r)
                outs.append((xo.cpu(), so.cpu(), {k: v.cpu() for k, v in idx.items() if hasattr(v, "cpu")}))
            for other in outs[1:]:
                assert torch.equal(outs[0][0], other[0]) and torch.equal(outs[0][1], other[1]), (name, layer)
                for k in ("node_idx", "unm_idx", "src_idx", "dst_idx"):
                    assert torch.equal(outs[0][2][k], other[2][k]), (name, layer, k)
            x, size = outs[0][0].float(), outs[0][1][..., None]
    finally:
        eng.set_option("tome_fused_ln", 1)
        eng.close()
