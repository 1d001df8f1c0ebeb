#!/usr/bin/env python3
"""What makes the wide epilogue's cycles depend on K?  (lab tool; needs `python -m aurora_amd.build --labs`)

On a QUIET chip (8 workgroups, one per XCD, 25 tiles each) the stamped kernel's epilogue takes ~21 k cycles per tile at K = 1280 and ~37 k at
K = 4096 (profiles/r04_gemm_segments_quiet_vs_full.log).  This probe varies one thing at a time: K, the output's row stride (N = 5120: a tile's
256 rows are 10 KB apart; N = 256: 512 B apart), stores on / off (lab 20), and prints the epilogue's cycles per tile.

    AURORA_HIP_SO=aurora_amd/libaurora_hip_labs.so python tools/gemm_lab/epi_probe.py
"""
import ctypes as C
import os
import sys

import numpy as np
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
os.environ.setdefault("AURORA_HIP_SO", os.path.join(ROOT, "aurora_amd", "libaurora_hip_labs.so"))


def main():
    from aurora_amd._lib import check
    from aurora_amd.engine import AuroraCapEngine, _rup
    eng = AuroraCapEngine({"vit": None, "llm": None}, {}, max_frames=1, max_batch=1)
    L = eng.L
    L.aur_lab_gemm_ts.restype = C.c_int
    L.aur_lab_gemm_ts.argtypes = [C.c_void_p, C.c_int]
    g = torch.Generator(device="cuda").manual_seed(0)
    ts = np.zeros(256 * 8 * 32, dtype=np.uint32)
    eng.set_option("gemm_mode", 2)

    def run(G, tiles_per_wg, K, N, lab):
        nbn = _rup(N, 256) // 256
        M = 256 * (G * tiles_per_wg // nbn)
        a = (torch.randn(M, K, generator=g, device="cuda") * 0.5).half()
        w = (torch.randn(N, K, generator=g, device="cuda") * 0.05).half()
        wp = eng.pack(w, _rup(N, 256), K)
        bias = torch.zeros(_rup(N, 256), device="cuda")
        c = torch.empty(M, N, dtype=torch.float16, device="cuda")
        eng.set_option("gemm_max_wgs", G)
        eng.set_option("gemm_lab", lab)
        st = eng._stream()
        call = lambda: check(eng.ctx, L.aur_linear(eng.ctx, a.data_ptr(), M, K, wp.data_ptr(), _rup(N, 256), N, bias.data_ptr(), 0, None, c.data_ptr(), st), "aur_linear")
        call()
        torch.cuda.synchronize()
        assert L.aur_lab_gemm_ts(ts.ctypes.data, 1) == 0
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        for _ in range(4):
            call()
        e1.record()
        torch.cuda.synchronize()
        us = e0.elapsed_time(e1) * 1e3 / 4
        assert L.aur_lab_gemm_ts(ts.ctypes.data, 1) == 0
        t = ts.reshape(256, 8, 32).astype(np.float64)
        tiles = t[:, :, 25].sum()
        epi, loop, blocks = t[:, :, 24].sum() / tiles, t[:, :, 21].sum() / tiles, t[:, :, 28].sum() / tiles
        ghz = (t[:, :, 22].sum() + t[:, :, 23].sum() + t[:, :, 24].sum()) / tiles / (us / tiles_per_wg * 1e3)
        print(f"  G {G:3d}  K {K:5d}  N {N:5d}  {'no stores' if lab == 20 else 'continuous' if lab == 23 else 'no operand DMA' if lab == 14 else 'lean epilogue' if lab == 26 else 'stores   '}  tile {us / tiles_per_wg:7.1f} us  K loop {loop:8.0f}  "
              f"epilogue {epi:7.0f} cycles (row blocks 1-7: {blocks:6.0f})  ~{ghz:4.2f} GHz", flush=True)
        eng.set_option("gemm_lab", 0)

    print("stamped kernel (lab 10; lab 20 = without the epilogue's stores), cycles per tile and wave")
    for G in (8, 256):
        for K in (256, 1280, 2560, 4096, 8192):
            run(G, 25, K, 5120, 10)
        for K in (1280, 4096):
            run(G, 25, K, 5120, 20)
        for K in (1280, 4096):
            run(G, 25, K, 256, 10)
        for K in (1280, 4096):           # lab 23: the continuous pipeline - no prologue burst in front of the epilogue
            run(G, 25, K, 5120, 23)
        for K in (256, 1280, 2560, 4096, 8192):     # lab 26: a compile-time activation = the product's 7.5 KB epilogue (the others: 78 KB)
            run(G, 25, K, 5120, 26)
        for K in (1280, 2560, 4096, 8192):     # lab 14: no operand DMA at all - the same MFMA bursts for the same time, nothing streamed
            run(G, 25, K, 5120, 14)
    eng.set_option("gemm_max_wgs", 0)
    eng.set_option("gemm_mode", 1)
    eng.close()


if __name__ == "__main__":
    main()
