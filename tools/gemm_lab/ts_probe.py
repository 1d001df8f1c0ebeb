#!/usr/bin/env python3
"""Where does a K-tile of the 256x256 GEMM go?  (lab tool; needs `python -m aurora_amd.build --labs`)

Runs gemm256.hip's lab instantiation 8 - the product kernel with s_memtime stamps around every segment of every phase, collected behind the
schedule's own lgkmcnt waits - on the path's shapes and prints, per phase, mean cycles per wave of
    load (LDS fragment reads + LDS-DMA issue) | barrier 1 | lgkmcnt wait | MFMA burst (16 MFMAs) | barrier 2
for the leading group (waves 0-3) and the trailing group (waves 4-7).  An ideal phase is 2 x 272 cycles (each group's burst hides the
other's load segment).

    AURORA_HIP_SO=aurora_amd/libaurora_hip_labs.so python tools/gemm_lab/ts_probe.py
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
    ts2 = np.zeros(256 * 8 * 4, dtype=np.uint32)
    L.aur_lab_gemm_ts2.restype = C.c_int
    L.aur_lab_gemm_ts2.argtypes = [C.c_void_p, C.c_int]
    STAMPED = (8, 9, 10, 13, 14, 16, 19, 20, 21, 23, 24)

    def run(name, M, K, N, lab, iters=8, with_bias=True):
        npad = _rup(N, 256)
        a = (torch.randn(_rup(M, 256), K, generator=g, device="cuda") * 0.5).half()      # whole 256-row tiles: lab 13 reads a K-tile-major image of them
        w = (torch.randn(N, K, generator=g, device="cuda") * 0.05).half()
        wp = eng.pack(w, npad, K)
        bias = torch.zeros(npad, device="cuda")
        c = torch.empty(M, N, dtype=torch.float16, device="cuda")
        eng.set_option("gemm_mode", 2)
        eng.set_option("gemm_lab", lab)
        st = eng._stream()
        call = lambda: check(eng.ctx, L.aur_linear(eng.ctx, a.data_ptr(), M, K, wp.data_ptr(), npad, N, bias.data_ptr() if with_bias else None, 0, None, c.data_ptr(), st), "aur_linear")
        for _ in range(2):
            call()
        torch.cuda.synchronize()
        assert L.aur_lab_gemm_ts(ts.ctypes.data, 1) == 0 and L.aur_lab_gemm_ts2(ts2.ctypes.data, 1) == 0
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        for _ in range(iters):
            call()
        e1.record()
        torch.cuda.synchronize()
        us = e0.elapsed_time(e1) * 1e3 / iters
        print(f"\n{name}: M {M} K {K} N {N} lab {lab}{'' if with_bias else ' (no bias)'}: {us:9.1f} us  {2.0 * M * N * K / us / 1e6:7.1f} TF/s", flush=True)
        if lab not in STAMPED:
            eng.set_option("gemm_lab", 0)
            return
        assert L.aur_lab_gemm_ts(ts.ctypes.data, 1) == 0 and L.aur_lab_gemm_ts2(ts2.ctypes.data, 1) == 0
        t = ts.reshape(256, 8, 32).astype(np.float64)
        t2 = ts2.reshape(256, 8, 4).astype(np.float64)
        tl = t[:, :, 25].sum()
        if tl > 0:
            f = [t[:, :, 30].sum() / tl, t[:, :, 31].sum() / tl] + [t2[:, :, k].sum() / tl for k in range(3)]
            print(f"    row block 0 in detail (cycles): bias adds + 4 ds_write issued {f[0]:6.0f} | 1st half's 2 ds_read back {f[1]:6.0f} | converted + stored {f[2]:6.0f} | "
                  f"2nd half's reads back {f[3]:6.0f} | stored {f[4]:6.0f}")
        for grp, ws in (("waves 0-3 (leading group)", slice(0, 4)), ("waves 4-7 (one barrier behind)", slice(4, 8))):
            x = t[:, ws].reshape(-1, 32)
            x = x[x[:, 20] > 0]
            n = x[:, 20].sum()
            seg = x[:, :20].sum(0).reshape(4, 5) / (n / 4.0)        # mean cycles per phase occurrence
            tot = x[:, 21].sum() / n
            print(f"  {grp}: mean cycles per phase {tot:7.1f} (ideal 2 x 272 = 544)")
            tiles = x[:, 25].sum()
            if tiles > 0:
                loop, headloop, pro, epi = x[:, 21].sum() / tiles, x[:, 22].sum() / tiles, x[:, 23].sum() / tiles, x[:, 24].sum() / tiles
                e = [x[:, k].sum() / tiles for k in (26, 27, 28, 29)]
                print(f"    epilogue split (cycles): entry -> bias / residual landed {e[0]:7.0f} | row block 0 {e[1]:6.0f} | row blocks 1-7 {e[2]:7.0f} | tail {e[3]:5.0f}")
                print(f"    per tile (cycles): head (wait for the prologue's K-tile 0 + stores + barriers) {headloop - loop:7.0f} | K loop {loop:8.0f} | "
                      f"next prologue issue {pro:6.0f} | epilogue {epi:7.0f} | total {headloop + pro + epi:8.0f}")
            print("    phase   load   bar1   lgkm   mfma   bar2    sum")
            for p in range(4):
                print(f"    {p}     " + " ".join(f"{v:6.0f}" for v in seg[p]) + f" {seg[p].sum():6.0f}")
        eng.set_option("gemm_lab", 0)

    print("labs: 0 product | 11 / 12 / 15 DMA schedule 1 / 2 / 4 (no stamps) | 8 / 9 / 10 / 16 stamps on schedule 0 / 1 / 2 / 4 | 13 stamps + A from a "
          "K-tile-major image | 14 stamps + no operand DMA | 17 / 18 start-up stagger over 4 / 8 slots per XCD, 19 = 17 with stamps | 22 / 23 continuous pipeline across tile boundaries (23 with stamps) | 20 / 21 stamps + no / half of the epilogue stores")
    shapes = [("llm gate/up, 4-clip prefill pass", 8576, 4096, 22016), ("llm down", 8576, 11008, 4096), ("llm qkv-shaped", 8576, 4096, 12288),
              ("vit fc1, t = 640 x 32 frames", 20480, 1280, 5120), ("vit fc2", 20480, 5120, 1280)]
    for rep in range(0):                                            # interleaved A/B/A/B of the un-stamped kernels
        for name, M, K, N in shapes:
            for lab in (0, 22):
                run(name, M, K, N, lab)
    for name, M, K, N in shapes[:1] + shapes[3:]:
        for lab in (10,):
            run(name, M, K, N, lab)
    # quiet chip against full chip: the same stamped kernel on 8 workgroups (one per XCD) and on 256, 25 tiles per workgroup
    # (profiles/r04_gemm_segments_quiet_vs_full.log: the same cycles per phase, 2.34 against 1.92 GHz)
    for G in (8, 256):
        eng.set_option("gemm_max_wgs", G)
        print(f"\n==== {G} workgroups x 25 tiles (M = {320 * G}, N = 5120)")
        for K in (1280, 4096):
            run(f"K sweep shape, {G} CUs", 320 * G, K, 5120, 10)
    eng.set_option("gemm_max_wgs", 0)
    eng.close()


if __name__ == "__main__":
    main()
