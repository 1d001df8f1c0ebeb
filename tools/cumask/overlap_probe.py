"""How do the HBM-bound decode step and the MFMA-bound front end share the chip?  (lab tool)
Times, on AuroraCap-7B shapes with B slots per bank: decode alone / front end alone under CU masks, then both at once.
  python tools/cumask/overlap_probe.py [B] [decode_steps]"""
import os
import sys
import time

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
from aurora_amd import synthetic as S                                            # noqa: E402
from aurora_amd.engine import AuroraCapEngine, _rup, tokens_at_layer, tome_r     # noqa: E402
from aurora_amd.streams import cu_masked_stream                                  # noqa: E402


def main():
    B = int(sys.argv[1]) if len(sys.argv) > 1 else 64
    steps = int(sys.argv[2]) if len(sys.argv) > 2 else 48
    F, N, G = 8, 256, 8
    NBANK = 2 if B <= 64 else 1                 # 128 slots: one bank fits; the front end then re-fills slots 0..7 (timing only)
    dev = "cuda:0"
    cfg = S.AURORACAP_7B
    v, l = cfg["vit"], cfg["llm"]
    weights = {"vit": S.vit_weights(v, device=dev), "projector": S.projector_weights(v["hidden_size"], l["hidden_size"], device=dev),
               "llm": S.llm_weights(l, device=dev)}
    t0tok = (v["image_size"] // v["patch_size"]) ** 2 + 1
    r = tome_r(v["image_size"], v["image_size"], v["patch_size"], 0.3, v["num_hidden_layers"])
    n_kept = tokens_at_layer(t0tok, r, v["num_hidden_layers"] - 1) - 1
    L0 = 30 + F * n_kept
    eng = AuroraCapEngine(cfg, weights, max_frames=G * F, max_batch=B, max_ctx=_rup(L0 + N, 64), max_new_tokens=N, num_banks=NBANK, device=dev)
    del weights
    torch.cuda.empty_cache()
    pixels = torch.cat([S.frames(F, b, v["image_size"], device=dev) for b in range(G)], 0)
    ids = [S.prompt_ids(F, b, 30, l["vocab_size"]) for b in range(G)]
    Mseq = _rup(L0, 32)
    emb = torch.zeros(G * Mseq, l["hidden_size"], dtype=torch.float16, device=dev)
    plans = [eng.splice_plan(ids[b], F, n_kept) for b in range(G)]

    def front(bank, slot0):
        eng.select_bank(bank)
        vis = eng.vit_encode(pixels, r)
        for j in range(G):
            eng.project_splice(vis[j * F:(j + 1) * F], plan=plans[j], out=emb[j * Mseq:(j + 1) * Mseq])
        eng.prefill_batch(slot0, G, emb, L0)

    # fill both banks (the same 8 clips in every group of slots: timing only)
    for bank in range(NBANK):
        eng.select_bank(bank)
        eng.begin_batch(B, N, None)
        for s0 in range(0, B, G):
            front(bank, s0)
    torch.cuda.synchronize()
    base = torch.cuda.current_stream()

    def timed(fn, stream):
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        stream.wait_stream(base)
        with torch.cuda.stream(stream):
            e0.record()
            fn()
            e1.record()
        return e0, e1

    def dec():
        eng.select_bank(0)
        eng.decode(steps)

    def fr():
        front(NBANK - 1, 0)

    def reset():
        eng.select_bank(0)
        eng.begin_batch(B, N, None)
        for s0 in range(0, B, G):
            front(0, s0)
        torch.cuda.synchronize()

    streams = {"all": base, "free": torch.cuda.Stream()}        # "free": an unmasked stream of its own (nothing waits on it)
    for k in (28, 24, 20, 16, 12, 8):
        streams[f"lo{k}"] = cu_masked_stream(k, device=dev)
        streams[f"hi{k}"] = cu_masked_stream(k, from_top=True, device=dev)
    print(f"B = {B} slots, {steps} decode steps, front end = {G} clips (ViT + splice + prefill)", flush=True)
    for name in ("all", "hi28", "hi24", "hi20", "hi16", "hi12"):
        e0, e1 = timed(dec, streams[name])
        torch.cuda.synchronize()
        print(f"decode alone  on {name:5s}: {e0.elapsed_time(e1) / steps:8.3f} ms/step", flush=True)
    for name in ("all", "lo24", "lo20", "lo16", "lo12", "lo8"):
        eng.set_option("gemm_max_wgs", 256 if name == "all" else 8 * int(name[2:]))
        e0, e1 = timed(fr, streams[name])
        torch.cuda.synchronize()
        print(f"front alone   on {name:5s}: {e0.elapsed_time(e1):8.1f} ms", flush=True)
    reset()
    for dn, fn in (("free", "all2"), ("free", "lo24"), ("free", "lo20"), ("free", "lo16"), ("free", "lo12"), ("hi16", "lo16"), ("hi20", "lo12"),
                   ("hi20", "lo16"), ("hi24", "lo16"), ("hi28", "lo16"), ("hi24", "lo12")):
        if fn == "all2":
            streams["all2"] = torch.cuda.Stream()
            eng.set_option("gemm_max_wgs", 256)
        else:
            eng.set_option("gemm_max_wgs", 8 * int(fn[2:]))
        t0 = time.perf_counter()
        d0, d1 = timed(dec, streams[dn])
        f0, f1 = timed(fr, streams[fn])
        torch.cuda.synchronize()
        wall = 1e3 * (time.perf_counter() - t0)
        print(f"together: decode on {dn:5s} {d0.elapsed_time(d1) / steps:8.3f} ms/step ({d0.elapsed_time(d1):7.1f} ms) | front on {fn:5s} "
              f"{f0.elapsed_time(f1):8.1f} ms | wall {wall:8.1f} ms", flush=True)
    eng.close()


if __name__ == "__main__":
    main()
