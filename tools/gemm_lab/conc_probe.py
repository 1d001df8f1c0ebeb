#!/usr/bin/env python3
"""Is the 256x256 GEMM's per-tile fixed cost (prologue + epilogue, ~17-20 us) a per-CU cost or a shared-resource cost?

The persistent kernel runs G workgroups (`gemm_max_wgs`), every workgroup walks 25 tiles of a problem sized to G (M = 320 G, N = 5120), at
K = 128 / 256 / 1280: time per tile = a + b * K / 64.  If `a` shrinks when fewer CUs run (all CUs reach their epilogue at the same moment:
256 x 128 KiB of stores in one burst), the fixed cost is the memory system's, not the CU's.

    python tools/gemm_lab/conc_probe.py
"""
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))


def main():
    from aurora_amd._lib import check
    from aurora_amd.engine import AuroraCapEngine, _rup
    eng = AuroraCapEngine({"vit": None, "llm": None}, {}, max_frames=1, max_batch=1)
    g = torch.Generator(device="cuda").manual_seed(0)
    L = eng.L
    eng.set_option("gemm_mode", 2)

    def run(M, K, N, iters=10, resid=False):
        npad = _rup(N, 256)
        a = (torch.randn(M, K, generator=g, device="cuda") * 0.5).half()
        w = (torch.randn(N, K, generator=g, device="cuda") * 0.05).half()
        wp = eng.pack(w, npad, K)
        bias = torch.zeros(npad, device="cuda")
        c = torch.empty(M, N, dtype=torch.float16, device="cuda")
        r = torch.randn(M, N, generator=g, device="cuda").half() if resid else None
        st = eng._stream()
        call = lambda: check(eng.ctx, L.aur_linear(eng.ctx, a.data_ptr(), M, K, wp.data_ptr(), npad, N, bias.data_ptr(), 0, r.data_ptr() if resid else None, c.data_ptr(), st), "aur_linear")
        for _ in range(3):
            call()
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        for _ in range(iters):
            call()
        e1.record()
        torch.cuda.synchronize()
        return e0.elapsed_time(e1) * 1e3 / iters

    print("G workgroups x 25 tiles each (N = 5120, M = 320 G): us per tile at K = 128 / 256 / 1280, fitted fixed cost a and slope b (us per 64 of K)")
    for resid in (False, True):
        print("with a residual operand" if resid else "bias only")
        for G in (8, 16, 32, 64, 128, 256):
            eng.set_option("gemm_max_wgs", G)
            M = 320 * G
            t = {K: run(M, K, 5120, resid=resid) / 25 for K in (128, 256, 1280)}
            b = (t[1280] - t[128]) / 18.0
            a = t[128] - 2 * b
            print(f"  G {G:4d}  M {M:6d}   {t[128]:7.2f} {t[256]:7.2f} {t[1280]:7.2f}   a {a:6.2f} us   b {b:5.2f} us", flush=True)
    eng.set_option("gemm_max_wgs", 0)
    eng.set_option("gemm_mode", 1)
    eng.close()


if __name__ == "__main__":
    main()
