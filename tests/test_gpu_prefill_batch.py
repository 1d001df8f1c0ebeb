"""GPU: batched prefill (several equal-length sequences in one pass of large-M GEMMs) must reproduce the
per-sequence prefill bit for bit - same logits, same KV pages, same generated tokens."""
import pytest
import torch

from tests.test_gpu_llm import LLM_CFGS, make_engine, padded

pytestmark = pytest.mark.gpu


def test_two_banks_overlapped_streams_equal_sequential():
    """Generation banks: batch A decodes on one stream while batch B is prefetched into the other bank on a second
    stream; both must produce exactly what they produce when run alone, one after the other."""
    from aurora_amd.engine import AuroraCapEngine
    from tests.util import rand_llm_weights
    cfg = LLM_CFGS["hd64"]
    w = rand_llm_weights(cfg, 14)
    gen = torch.Generator().manual_seed(41)
    A = [torch.randn(L, cfg["hidden_size"], generator=gen).half().float() for L in (200, 200, 200)]
    Bq = [torch.randn(L, cfg["hidden_size"], generator=gen).half().float() for L in (333, 333, 333)]
    eng = AuroraCapEngine({"vit": None, "llm": cfg}, {"llm": w}, max_frames=1, max_batch=3, max_ctx=512, max_new_tokens=40,
                          num_banks=2)
    try:
        def big(xs):
            return torch.cat([padded(e) for e in xs], 0).contiguous()
        ref = {}
        for name, xs in (("A", A), ("B", Bq)):                      # sequential reference, bank 0
            eng.select_bank(0)
            eng.begin_batch(3, 40, None)
            eng.prefill_batch(0, 3, big(xs), xs[0].shape[0])
            eng.decode(39)
            ref[name] = eng.outputs()
        for rep in range(3):                                         # overlapped: A decodes in bank 0 || B front end in bank 1
            sD = torch.cuda.current_stream()
            sP = torch.cuda.Stream()
            sP.wait_stream(sD)
            eng.select_bank(0)
            eng.begin_batch(3, 40, None)
            eng.prefill_batch(0, 3, big(A), 200)
            with torch.cuda.stream(sP):
                eng.select_bank(1)
                eng.begin_batch(3, 40, None)
                bb = big(Bq)
                eng.prefill_batch(0, 3, bb, 333)
                evp = torch.cuda.Event()
                evp.record(sP)
            eng.select_bank(0)
            eng.decode(39)
            outA = eng.outputs()
            sD.wait_event(evp)
            eng.select_bank(1)
            eng.decode(39)
            outB = eng.outputs()
            assert outA == ref["A"] and outB == ref["B"], rep
        with pytest.raises(Exception):
            eng.select_bank(2)
    finally:
        eng.select_bank(0)
        eng.close()


def test_batch_of_52_four_column_groups_equals_single():
    """Up to 64 slots = four MFMA column groups in the decode GEMVs; ragged prompts, EOS stopping per slot."""
    cfg = LLM_CFGS["hd64"]
    gen = torch.Generator().manual_seed(33)
    lens = [40 + (7 * i) % 90 for i in range(52)]
    embs = [torch.randn(L, cfg["hidden_size"], generator=gen).half().float() for L in lens]
    eng, w = make_engine(cfg, 13, max_batch=52, use_graph=True, max_ctx=256, max_new=10)
    try:
        together = eng.generate([padded(e) for e in embs], lens, 10, eos_id=None)
        for b in (0, 15, 16, 31, 32, 33, 47, 48, 51):
            alone = eng.generate([padded(embs[b])], [lens[b]], 10, eos_id=None)[0]
            assert together[b] == alone, b
        eos = together[40][4]
        stopped = eng.generate([padded(e) for e in embs], lens, 10, eos_id=eos)
        for b in range(52):
            want = together[b][: together[b].index(eos) + 1] if eos in together[b] else together[b]
            assert stopped[b] == want, b
    finally:
        eng.close()
    from aurora_amd._lib import AuroraHipError
    with pytest.raises(AuroraHipError):
        make_engine(cfg, 13, max_batch=129)                 # the slot cap (AUR_MAX_BATCH: 8 MFMA column groups)


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


@pytest.mark.parametrize("use_graph,masked", [(True, False), (False, False), (True, True)])
def test_staged_prefill_on_a_second_stream_while_all_slots_decode(use_graph, masked):
    """aur_llm_prefill_stage writes spare KV sequences on a second (optionally CU-masked) stream while every slot keeps
    decoding; aur_llm_prefill_commit hands the pages to the slots between two decode calls.  The untouched slots must
    generate exactly what they generate without the interleaved front end, and the re-filled slots exactly what their
    prompts generate alone - twice over, so that the second round runs on the exchanged page-table rows."""
    from aurora_amd.engine import AuroraCapEngine
    from tests.util import rand_llm_weights
    cfg = LLM_CFGS["hd64"]
    w = rand_llm_weights(cfg, 17)
    gen = torch.Generator().manual_seed(51)
    first = [torch.randn(100, cfg["hidden_size"], generator=gen).half().float() for _ in range(4)]
    nxt = [torch.randn(120, cfg["hidden_size"], generator=gen).half().float() for _ in range(2)]
    third = [torch.randn(77, cfg["hidden_size"], generator=gen).half().float() for _ in range(2)]
    NEW = 40
    eng = AuroraCapEngine({"vit": None, "llm": cfg}, {"llm": w}, max_frames=1, max_batch=4, max_ctx=512, max_new_tokens=NEW,
                          use_graph=use_graph, spare_slots=2)
    try:
        def big(xs):
            return torch.cat([padded(e) for e in xs], 0).contiguous()
        alone = {}
        for name, xs in (("first", first), ("nxt", nxt), ("third", third)):
            alone[name] = eng.generate([padded(e) for e in xs], [xs[0].shape[0]] * len(xs), NEW, eos_id=None)
        sD = torch.cuda.current_stream()
        if masked:
            from aurora_amd.streams import cu_masked_stream
            sF = cu_masked_stream(12)
        else:
            sF = torch.cuda.Stream()
        eng.begin_batch(4, NEW, None)
        eng.prefill_batch(0, 4, big(first), 100)
        eng.decode(7)                                               # every slot has 8 tokens
        # front end for slots 1, 2 into the spare sequences 4, 5 - concurrently with 9 more decode steps of ALL slots
        emb2 = big(nxt)
        sF.wait_stream(sD)
        with torch.cuda.stream(sF):
            eng.prefill_stage(4, 2, emb2, 120)
            ev = torch.cuda.Event()
            ev.record(sF)
        eng.decode(9)                                               # 17 tokens
        sD.wait_event(ev)
        eng.prefill_commit(1, 2, 4, emb2, 120)                      # slots 1, 2 restart with 1 token each
        evc = torch.cuda.Event()
        evc.record(sD)
        # second round: slots 2, 3 <- `third`, staged into sequences 4, 5 again (they now own the old pages of slots 1, 2)
        emb3 = big(third)
        sF.wait_event(evc)
        with torch.cuda.stream(sF):
            eng.prefill_stage(4, 2, emb3, 77)
            ev2 = torch.cuda.Event()
            ev2.record(sF)
        eng.decode(10)                                              # slot 0: 27 tokens, slots 1, 2: 11 tokens, slot 3: 27
        out_mid = eng.outputs()
        sD.wait_event(ev2)
        eng.prefill_commit(2, 2, 4, emb3, 77)
        eng.decode(12)                                              # slot 0: 39, slot 1: 23, slots 2, 3: 13
        out = eng.outputs()
        assert out_mid[0] == alone["first"][0][:27] and out_mid[3] == alone["first"][3][:27]
        assert out_mid[1] == alone["nxt"][0][:11] and out_mid[2] == alone["nxt"][1][:11]
        assert out[0] == alone["first"][0][:39]
        assert out[1] == alone["nxt"][0][:23]
        assert out[2] == alone["third"][0][:13] and out[3] == alone["third"][1][:13]
        from aurora_amd._lib import AuroraHipError
        with pytest.raises(AuroraHipError):
            eng.prefill_stage(5, 2, emb3, 77)                       # sequences [5, 7) exceed max_batch + spare_slots
        with pytest.raises(AuroraHipError):
            eng.prefill_commit(3, 2, 4, emb3, 77)                   # slots [3, 5) exceed the batch
    finally:
        eng.close()
