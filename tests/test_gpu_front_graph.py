"""GPU: a front end captured into a hipGraph (aur_graph_begin / _end / _launch, engine.FrontEndGraph - the per-shape-bucket capture of
the reference tree's serving engine, src/sglang/python/sglang/srt/model_executor/cuda_graph_runner.py:163-279) must produce, replay after
replay and clip after clip, bitwise what the eager call sequence produces; a broken capture must fail loudly and leave the ctx usable;
the in-loop stamps of the decode step must not change a token."""
import numpy as np
import pytest
import torch

from aurora_amd import _lib
from tests.test_gpu_caption_batch import VCFG, weights
from tests.test_gpu_llm import LLM_CFGS

pytestmark = pytest.mark.gpu


def build(max_batch, spare, max_new=12):
    from aurora_amd.engine import AuroraCapEngine
    return AuroraCapEngine({"vit": VCFG, "llm": LLM_CFGS["hd32"]}, weights(), max_frames=6, max_batch=max_batch, max_ctx=512,
                           max_new_tokens=max_new, spare_slots=spare)


def group_clips(n, frames, seed):
    gen = torch.Generator().manual_seed(seed)
    return [(torch.randn(frames, 3, 56, 56, generator=gen).half().cuda(), [1, 20 + i] + [-200, 30 + i] * frames + [40 + i, 41]) for i in range(n)]


def test_captured_front_end_is_bitwise_the_eager_one():
    from aurora_amd.engine import FrontEndGraph
    G, F, B, N = 2, 3, 4, 10
    eng = build(B, G, N)
    try:
        r = eng.tome_r(0.5)
        cs = group_clips(6, F, 11)
        alone = [eng.caption_ids(px, ids, 0.5, N, eos_id=None) for px, ids in cs]
        n_kept = eng.vit_encode(cs[0][0], r).shape[1]
        plans = [eng.splice_plan(ids, F, n_kept) for _, ids in cs]
        L = plans[0]["seq_len"]
        # eager reference of the staged group: hidden states left in `embeds` + first-token logits after the commit
        eng.begin_batch(B, N, None)
        for s in range(B):
            eng.slot_retire(s)
        fg = FrontEndGraph(eng, G, F, 56, 56, r, plans[0], seq0=B)
        assert fg.nodes > 20, fg.nodes
        mseq = fg.mseq
        for rep, g0 in enumerate((0, 2, 4, 0)):                  # the graph is replayed with different clips in its static buffers
            emb_e = torch.zeros(G * mseq, eng.l["hidden_size"], dtype=torch.float16, device="cuda")
            vis = eng.vit_encode(torch.cat([cs[g0 + j][0] for j in range(G)]), r)
            for j in range(G):
                eng.project_splice(vis[j * F:(j + 1) * F], plan=plans[g0 + j], out=emb_e[j * mseq:(j + 1) * mseq])
            eng.prefill_stage(B, G, emb_e, L)
            torch.cuda.synchronize()
            want_hidden = emb_e.clone()
            fg.load(torch.cat([cs[g0 + j][0] for j in range(G)]), plans[g0:g0 + G])
            fg.launch()
            torch.cuda.synchronize()
            assert torch.equal(fg.embeds[:G * mseq], want_hidden), rep
            # commit the replayed group into slots [0, G) and decode: each clip's ids are the ids it gets alone
            eng.prefill_commit(0, G, B, fg.embeds, L)
            eng.decode(N - 1)
            out = eng.outputs()
            for j in range(G):
                assert out[j] == alone[g0 + j], (rep, j)
        # stacked plan arrays instead of plan dicts
        vr = torch.stack([plans[4]["vis_rows"], plans[5]["vis_rows"]])
        ti = torch.stack([plans[4]["text_ids"], plans[5]["text_ids"]])
        tr = torch.stack([plans[4]["text_rows"], plans[5]["text_rows"]])
        fg.load(torch.cat([cs[4][0], cs[5][0]]), vis_rows=vr, text_ids=ti, text_rows=tr)
        fg.launch()
        eng.prefill_commit(2, G, B, fg.embeds, L)
        eng.decode(N - 1)
        out = eng.outputs()
        assert out[2] == alone[4] and out[3] == alone[5]
        with pytest.raises(ValueError):
            fg.load(torch.cat([cs[0][0], cs[1][0]]), [plans[0], eng.splice_plan([1, 2, 3] + cs[1][1], F, n_kept)])
        fg.close()
        with pytest.raises(_lib.AuroraHipError):
            eng.graph_launch(fg.gid or 1)
    finally:
        eng.close()


def test_capture_on_a_cu_masked_stream_and_batch_prefill():
    """The serving loop captures and replays on the front-end stream (hipExtStreamCreateWithCUMask); the `batch` flavour of the graph
    (prefill straight into decode slots, first tokens included) equals caption_batch."""
    from aurora_amd.engine import FrontEndGraph
    from aurora_amd.streams import cu_masked_stream
    G, F, N = 2, 2, 8
    eng = build(G, 0, N)
    try:
        r = eng.tome_r(0.5)
        cs = group_clips(2, F, 5)
        want = eng.caption_batch(cs, 0.5, N, eos_id=None)
        n_kept = eng.vit_encode(cs[0][0], r).shape[1]
        plans = [eng.splice_plan(ids, F, n_kept) for _, ids in cs]
        sF = cu_masked_stream(16, device="cuda:0")
        torch.cuda.synchronize()
        eng.begin_batch(G, N, None)
        torch.cuda.synchronize()
        with torch.cuda.stream(sF):
            fg = FrontEndGraph(eng, G, F, 56, 56, r, plans[0], seq0=0, prefill="batch", slot0=0)
            for s in range(G):
                eng.slot_reset(s)
            fg.load(torch.cat([c[0] for c in cs]), plans)
            fg.launch()
            eng.decode(N - 1)
            got = eng.outputs()
        assert got == want
    finally:
        eng.close()


def test_broken_capture_fails_loudly_and_the_ctx_survives():
    eng = build(2, 0)
    try:
        px, ids = group_clips(1, 2, 3)[0]
        want = eng.caption_ids(px, ids, 0.5, 6, eos_id=None)
        with pytest.raises(_lib.AuroraHipError):
            eng.graph_capture(lambda: eng.decode(1))              # the decode step replays its own graph: refused inside a capture
        with pytest.raises(_lib.AuroraHipError):
            eng.graph_capture(lambda: eng.slot_state())           # synchronises: the runtime invalidates the capture
        with pytest.raises(_lib.AuroraHipError):
            eng.graph_launch(1)                                   # nothing was instantiated
        torch.cuda.synchronize()
        assert eng.caption_ids(px, ids, 0.5, 6, eos_id=None) == want
        gid, nodes = eng.graph_capture(lambda: eng.slot_retire(0))
        assert gid >= 1 and nodes >= 1
        eng.graph_launch(gid)
        eng.graph_destroy(gid)
    finally:
        eng.close()


def test_decode_stamps_time_the_step_without_changing_a_token():
    eng = build(2, 0, 12)
    try:
        cs = group_clips(2, 2, 7)
        want = eng.caption_batch(cs, 0.5, 12, eos_id=None)
        eng.set_option("decode_stamp_layer", 1)
        got = eng.caption_batch(cs, 0.5, 12, eos_id=None)
        us, total = eng.decode_stamps()
        assert got == want
        assert total >= 11 and len(us) == min(total, 4096)
        assert (us > 0).all() and (us < 5e4).all(), us            # a tiny attention launch + two boundaries: microseconds
        eng.set_option("decode_stamp_layer", -1)
        assert eng.caption_batch(cs, 0.5, 12, eos_id=None) == want
        assert eng.decode_stamps()[1] == total                    # no further stamps
        with pytest.raises(_lib.AuroraHipError):
            eng.set_option("decode_stamp_layer", 99)
    finally:
        eng.close()


def test_caption_stream_replays_one_graph_per_shape_bucket():
    """The product's overlapped stream (engine.caption_stream on an engine with spare KV sequences) submits each clip's front end as one
    graph replay; buckets are captured on first use, evicted least-recently-used beyond `front_graph_cap`; ids equal the eager stream's
    and each clip's own."""
    from tests.test_gpu_caption_batch import clips
    eng = build(3, 2, 16)
    try:
        cs = clips(12, 21)
        alone = [eng.caption_ids(px, ids, 0.5, 16, eos_id=None) for px, ids in cs]
        eager = dict(eng.caption_stream(cs, 0.5, 16, eos_id=None, check_every=4, front_graph=False))
        assert not eng._front_graphs
        got = dict(eng.caption_stream(cs, 0.5, 16, eos_id=None, check_every=4))
        assert got == eager == {i: alone[i] for i in range(12)}
        n_buckets = len(eng._front_graphs)
        assert 2 <= n_buckets <= eng.front_graph_cap
        again = dict(eng.caption_stream(cs, 0.5, 16, eos_id=None, check_every=5))      # every bucket is replayed, none re-captured
        assert again == eager and len(eng._front_graphs) == n_buckets
        eng.front_graph_cap = 2                                    # eviction on nearly every clip: still the same ids
        assert dict(eng.caption_stream(cs, 0.5, 16, eos_id=None, check_every=4)) == eager
        assert len(eng._front_graphs) <= n_buckets
    finally:
        eng.close()
