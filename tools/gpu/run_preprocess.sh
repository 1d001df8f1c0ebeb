#!/bin/bash
# GPU: input-stage parity tests + kernel timing (HIP events) for typical clip shapes
mkdir -p gpurun_out
timeout 900 python -m pytest tests/test_gpu_preprocess.py -x -q 2>&1 | tail -5 | tee gpurun_out/preprocess_tests.log
timeout 600 python - <<'PY' 2>&1 | tee gpurun_out/preprocess_bench.log
import torch, numpy as np
from aurora_amd.preprocess import FramePreprocessor
pre = FramePreprocessor()
for (f, h, w) in [(8, 480, 640), (8, 720, 1280), (8, 1080, 1920), (16, 720, 1280), (64, 720, 1280), (8, 2160, 3840)]:
    x = torch.randint(0, 256, (f, h, w, 3), dtype=torch.uint8, device="cuda")
    out = torch.empty(f, 3, 378, 378, dtype=torch.float16, device="cuda")
    for _ in range(3): pre(x, out=out)
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    n = 50
    e0.record()
    for _ in range(n): pre(x, out=out)
    e1.record(); torch.cuda.synchronize()
    us = e0.elapsed_time(e1) * 1e3 / n
    plan = pre.plan(h, w).cpu().numpy()
    left, ksh = int(plan[3]), int(plan[4])
    cols = min(w, int(378 * max(w / plan[1], 1.0)) + 2 * ksh)          # input columns actually touched
    alg = f * (h * cols * 3 + 2 * h * 378 * 3 + 378 * 378 * 3 * 2)
    print(f"f={f} {h}x{w}: {us:8.1f} us/call  {us / f:7.2f} us/frame  algorithmic {alg / 1e6:7.1f} MB -> {alg / us / 1e6:6.2f} TB/s")
PY
