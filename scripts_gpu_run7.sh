#!/bin/bash
mkdir -p gpurun_out
echo "=== tests" | tee gpurun_out/run7.log
timeout 900 python -m pytest tests/test_gpu_llm.py tests/test_gpu_kernels.py -q -m gpu --no-header -p no:cacheprovider -x 2>&1 | tail -4 | tee -a gpurun_out/run7.log
echo "=== microbench B=8 fused vs unfused" | tee -a gpurun_out/run7.log
timeout 900 python - <<'PY' 2>&1 | grep -v amdgpu.ids | tee -a gpurun_out/run7.log
import sys, torch
sys.path.insert(0, '.')
from aurora_amd import synthetic as S
from aurora_amd.engine import AuroraCapEngine, _rup
l = S.VICUNA_7B_16K
for B in (8, 16):
    eng = AuroraCapEngine({"vit": None, "llm": l}, {"llm": S.llm_weights(l)}, max_frames=1, max_batch=B, max_ctx=2432, max_new_tokens=256)
    eng.begin_batch(B, 256, None)
    for b in range(B):
        eng.prefill(b, (torch.randn(2144, 4096, device="cuda") * 0.02).half(), 2142)
    for fuse in (0, 1):
        eng.set_option("fuse_norm", fuse)
        print(B, "fuse", fuse, {k: round(eng.microbench(k, 320), 2) for k in ("dec_norm", "dec_qkv", "dec_gateup")}, flush=True)
    eng.close(); del eng; torch.cuda.empty_cache()
PY
for b in 16; do
  echo "=== bench B=$b" | tee -a gpurun_out/run7.log
  timeout 1200 python bench.py --steps 2 --warmup 1 --batch $b --no-cpu-baseline 2>&1 | tail -1 | python -c "
import sys, json
d = json.loads(sys.stdin.read())
print({k: d[k] for k in ('value','ms_per_step','p50_ttft_ms','ttft_ms_single_clip','stage_ms_instrumented_step')}, d.get('roofline',{}).get('achieved'), d.get('roofline',{}).get('avg_launch_us'))" | tee -a gpurun_out/run7.log
done
