#!/usr/bin/env python3
"""Cache policy of the 256x256 GEMM's output stores (lab tool; needs `python -m aurora_amd.build --labs`).

Every round of tiles writes 256 x 128 KiB = the capacity of the eight L2s.  Lab 25 issues the wide epilogue's stores with the default policy,
lab 12 is the same generic-activation kernel with the product's non-temporal stores; the path's shapes, A/B/A/B.  (The first run of round 4
also had sc1 and sc0 sc1 buffer stores: profiles/r04_gemm_store_policy.log.)

    AURORA_HIP_SO=aurora_amd/libaurora_hip_labs.so python tools/gemm_lab/store_probe.py
"""
import os
import sys

import torch

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
os.environ.setdefault("AURORA_HIP_SO", os.path.join(ROOT, "aurora_amd", "libaurora_hip_labs.so"))


def main():
    from aurora_amd._lib import AUR_ACT_QUICK_GELU, check
    from aurora_amd.engine import AuroraCapEngine, _rup
    eng = AuroraCapEngine({"vit": None, "llm": None}, {}, max_frames=1, max_batch=1)
    g = torch.Generator(device="cuda").manual_seed(0)
    L = eng.L
    eng.set_option("gemm_mode", 2)
    shapes = [("vit fc1  t640", 20480, 1280, 5120, AUR_ACT_QUICK_GELU, False), ("vit fc2  t640", 20480, 5120, 1280, 0, True),
              ("vit out  t640", 20480, 1280, 1280, 0, True), ("llm gateup", 8576, 4096, 22016, 0, False),
              ("llm down", 8576, 11008, 4096, 0, True), ("llm o", 8576, 4096, 4096, 0, True),
              ("K sweep 128", 81920, 128, 5120, 0, False), ("K sweep 1280", 81920, 1280, 5120, 0, False)]
    labs = ((25, "default"), (12, "nt"))
    print(f"{'shape (M, K, N)':36s} " + " ".join(f"{n:>10s}" for _, n in labs) + "   (us, two passes)")
    for name, M, K, N, act, resid in shapes:
        npad = _rup(N, 256)
        a = (torch.randn(M, K, generator=g, device="cuda") * 0.5).half()
        w = (torch.randn(N, K, generator=g, device="cuda") * 0.05).half()
        wp = eng.pack(w, npad, K)
        bias = torch.zeros(npad, device="cuda")
        c = torch.empty(M, N, dtype=torch.float16, device="cuda")
        r = torch.randn(M, N, generator=g, device="cuda").half() if resid else None
        st = eng._stream()
        call = lambda: check(eng.ctx, L.aur_linear(eng.ctx, a.data_ptr(), M, K, wp.data_ptr(), npad, N, bias.data_ptr(), act, r.data_ptr() if resid else None, c.data_ptr(), st), "aur_linear")
        ref = None
        for rep in range(2):
            row = []
            for lab, _ in labs:
                eng.set_option("gemm_lab", lab)
                for _ in range(3):
                    call()
                e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
                e0.record()
                for _ in range(10):
                    call()
                e1.record()
                torch.cuda.synchronize()
                row.append(e0.elapsed_time(e1) * 100.0)
                if ref is None:
                    ref = c.clone()
                assert torch.equal(c, ref), f"lab {lab} changed the result"
            print(f"{name:14s} {str((M, K, N)):21s} " + " ".join(f"{v:10.1f}" for v in row), flush=True)
    eng.set_option("gemm_lab", 0)
    eng.set_option("gemm_mode", 1)
    eng.close()


if __name__ == "__main__":
    main()
