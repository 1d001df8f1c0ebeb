#!/bin/bash
mkdir -p gpurun_out
timeout 900 python - <<'PY' 2>&1 | grep -v amdgpu.ids | tee gpurun_out/run13.log
import sys, torch
sys.path.insert(0, '.')
from aurora_amd import synthetic as S
from aurora_amd.engine import AuroraCapEngine
l = S.VICUNA_7B_16K
for B in (8, 32):
    eng = AuroraCapEngine({"vit": None, "llm": l}, {"llm": S.llm_weights(l)}, max_frames=1, max_batch=B, max_ctx=2432, max_new_tokens=256)
    eng.begin_batch(B, 256, None)
    emb = (torch.randn(B * 2144, 4096, device="cuda") * 0.02).half()
    eng.prefill_batch(0, B, emb, 2142)
    for nw in (4, 8, 16):
        eng.set_option("norm_waves", nw)
        print("B", B, "norm waves", nw, round(eng.microbench("dec_norm", 640), 2), "us", flush=True)
    eng.close(); del eng; torch.cuda.empty_cache()
PY
