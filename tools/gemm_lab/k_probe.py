#!/usr/bin/env python3
"""One EPI_ROW GEMM shape in a loop (for rocprofv3 --pmc passes over gemm256_kernel):  python tools/gemm_lab/k_probe.py M K N [iters]
K = 128 makes a tile ~85 % epilogue: the counters then describe the epilogue; K = 5120 the K loop."""
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))


def main():
    from aurora_amd._lib import check
    from aurora_amd.engine import AuroraCapEngine, _rup
    M, K, N = (int(x) for x in sys.argv[1:4])
    iters = int(sys.argv[4]) if len(sys.argv) > 4 else 10
    eng = AuroraCapEngine({"vit": None, "llm": None}, {}, max_frames=1, max_batch=1)
    g = torch.Generator(device="cuda").manual_seed(0)
    npad = _rup(N, 256)
    a = (torch.randn(M, K, generator=g, device="cuda") * 0.5).half()
    w = (torch.randn(N, K, generator=g, device="cuda") * 0.05).half()
    wp = eng.pack(w, npad, K)
    c = torch.empty(M, N, dtype=torch.float16, device="cuda")
    eng.set_option("gemm_mode", 2)
    for _ in range(iters):
        check(eng.ctx, eng.L.aur_linear(eng.ctx, a.data_ptr(), M, K, wp.data_ptr(), npad, N, None, 0, None, c.data_ptr(), eng._stream()), "aur_linear")
    torch.cuda.synchronize()
    eng.close()


if __name__ == "__main__":
    main()
