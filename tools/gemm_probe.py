#!/usr/bin/env python3
"""Time aur_linear (EPI_ROW GEMM) on the path's shapes with both tile kernels; fit the per-tile fixed cost from a K sweep.

    python tools/gemm_probe.py
"""
import ctypes as C
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))


def main():
    from aurora_amd._lib import AUR_ACT_QUICK_GELU, check
    from aurora_amd.engine import AuroraCapEngine, _rup
    eng = AuroraCapEngine({"vit": None, "llm": None}, {}, max_frames=1, max_batch=1)
    g = torch.Generator(device="cuda").manual_seed(0)
    L = eng.L

    def run(M, K, N, mode, act=0, resid=False, iters=20):
        npad = _rup(N, 256)
        a = (torch.randn(M, K, generator=g, device="cuda") * 0.5).half()
        w = (torch.randn(N, K, generator=g, device="cuda") * 0.05).half()
        wp = eng.pack(w, npad, K)
        bias = torch.zeros(npad, device="cuda")
        c = torch.empty(M, N, dtype=torch.float16, device="cuda")
        r = torch.randn(M, N, generator=g, device="cuda").half() if resid else None
        eng.set_option("gemm_mode", mode)
        st = eng._stream()
        call = lambda: check(eng.ctx, L.aur_linear(eng.ctx, a.data_ptr(), M, K, wp.data_ptr(), npad, N, bias.data_ptr(), act,
                                                   r.data_ptr() if r is not None else None, c.data_ptr(), st), "aur_linear")
        for _ in range(3):
            call()
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        for _ in range(iters):
            call()
        e1.record()
        torch.cuda.synchronize()
        us = e0.elapsed_time(e1) * 1e3 / iters
        return us, 2.0 * M * N * K / us / 1e6

    print(f"{'shape (M, K, N)':34s} {'kernel':8s} {'us':>9s} {'TF/s':>8s}")
    shapes = [("vit fc1  t640", 81920, 1280, 5120, AUR_ACT_QUICK_GELU, False), ("vit fc2  t640", 81920, 5120, 1280, 0, True),
              ("vit out  t640", 81920, 1280, 1280, 0, True), ("vit fc1  t288", 36864, 1280, 5120, AUR_ACT_QUICK_GELU, False),
              ("vit fc2  t288", 36864, 5120, 1280, 0, True), ("llm gateup", 17152, 4096, 22016, 0, False),
              ("llm down", 17152, 11008, 4096, 0, True), ("llm o", 17152, 4096, 4096, 0, True)]
    for name, M, K, N, act, res in shapes:
        for mode, tag in ((0, "128x128"), (2, "256x256")):
            us, tf = run(M, K, N, mode, act, res)
            print(f"{name:14s} {str((M, K, N)):20s} {tag:8s} {us:9.1f} {tf:8.1f}", flush=True)
    print("K sweep at M = 81920, N = 5120 (6400 tiles of 256x256 = 25 per CU): time per tile = a + b * K/64")
    for K in (128, 256, 640, 1280, 2560, 5120):
        us, tf = run(81920, K, 5120, 2, 0, False, iters=10)
        print(f"  K {K:5d}  {us:9.1f} us  {tf:8.1f} TF/s   per tile {us / 25:7.2f} us", flush=True)
    eng.set_option("gemm_mode", 1)
    eng.close()


if __name__ == "__main__":
    main()
