#!/usr/bin/env python3
"""One clip's ViT-H + ToMe pass, repeated (for rocprofv3 --kernel-trace --stats: per-kernel time of the single-clip front end).
    python tools/vit_single_clip.py [frames=8] [reps=20] [ratio=0.3]"""
import os
import sys

import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from aurora_amd import synthetic as S  # noqa: E402
from aurora_amd.engine import AuroraCapEngine, tome_r  # noqa: E402

F = int(sys.argv[1]) if len(sys.argv) > 1 else 8
reps = int(sys.argv[2]) if len(sys.argv) > 2 else 20
ratio = float(sys.argv[3]) if len(sys.argv) > 3 else 0.3
v = S.AURORACAP_7B["vit"]
eng = AuroraCapEngine({"vit": v, "llm": None}, {"vit": S.vit_weights(v)}, max_frames=F, max_batch=1, max_ctx=128, max_new_tokens=8)
px = S.frames(F, 0, v["image_size"], device="cuda")
r = tome_r(v["image_size"], v["image_size"], v["patch_size"], ratio, v["num_hidden_layers"])
for _ in range(3):
    eng.vit_encode(px, r)
torch.cuda.synchronize()
e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
e0.record()
for _ in range(reps):
    eng.vit_encode(px, r)
e1.record()
torch.cuda.synchronize()
print(f"ViT-H + ToMe, {F} frames, r = {r}: {e0.elapsed_time(e1) / reps:.3f} ms per pass")
eng.close()
