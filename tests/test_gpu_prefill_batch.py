"""GPU: batched prefill (several equal-length sequences in one pass of large-M GEMMs) must reproduce the
per-sequence prefill bit for bit - same logits, same KV pages, same generated tokens."""
import pytest
import torch

from tests.test_gpu_llm import LLM_CFGS, make_engine, padded

pytestmark = pytest.mark.gpu


def test_batch_of_20_two_column_groups_equals_single():
    """More than 16 slots use two MFMA column groups in the decode GEMVs; every slot must still generate exactly
    what it generates alone (batch-invariant arithmetic), including ragged prompt lengths."""
    cfg = LLM_CFGS["hd32"]
    gen = torch.Generator().manual_seed(31)
    lens = [33 + 3 * i for i in range(20)]
    embs = [torch.randn(L, cfg["hidden_size"], generator=gen).half().float() for L in lens]
    eng, w = make_engine(cfg, 12, max_batch=20, use_graph=True, max_ctx=256, max_new=12)
    try:
        together = eng.generate([padded(e) for e in embs], lens, 12, eos_id=None)
        for b in (0, 7, 15, 16, 19):
            alone = eng.generate([padded(embs[b])], [lens[b]], 12, eos_id=None)[0]
            assert together[b] == alone, b
    finally:
        eng.close()


@pytest.mark.parametrize("name,L", [("hd128", 97), ("hd32", 40), ("hd64", 300)])
def test_prefill_batch_equals_per_slot(name, L):
    cfg = LLM_CFGS[name]
    nseq = 3
    gen = torch.Generator().manual_seed(21)
    embs = [torch.randn(L, cfg["hidden_size"], generator=gen).half().float() for _ in range(nseq)]
    eng, w = make_engine(cfg, 11, max_batch=nseq, use_graph=True, max_ctx=512)
    try:
        # per slot
        eng.begin_batch(nseq, 10, None)
        for b in range(nseq):
            eng.prefill(b, padded(embs[b]), L)
        lg1 = eng.logits().clone()
        eng.decode(9)
        out1 = eng.outputs()
        # one pass
        big = torch.cat([padded(e) for e in embs], 0).contiguous()
        eng.begin_batch(nseq, 10, None)
        eng.prefill_batch(0, nseq, big, L)
        lg2 = eng.logits().clone()
        eng.decode(9)
        out2 = eng.outputs()
        assert torch.equal(lg1, lg2)
        assert out1 == out2
        # and the 256x256 kernel on the batched shape
        eng.set_option("gemm_mode", 2)
        big = torch.cat([padded(e) for e in embs], 0).contiguous()
        eng.begin_batch(nseq, 10, None)
        eng.prefill_batch(0, nseq, big, L)
        eng.decode(9)
        assert eng.outputs() == out1
    finally:
        eng.set_option("gemm_mode", 1)
        eng.close()
