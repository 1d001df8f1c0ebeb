#!/bin/bash
# gemm256 correctness + speed
mkdir -p gpurun_out
echo "=== tests" | tee gpurun_out/run8.log
for f in tests/test_gpu_kernels.py tests/test_gpu_vit.py tests/test_gpu_llm.py; do
  timeout 900 python -m pytest $f -q -m gpu --no-header -p no:cacheprovider 2>&1 | tail -12 | tee -a gpurun_out/run8.log
done
echo "=== microbench B=8 quick" | tee -a gpurun_out/run8.log
timeout 900 python tools/microbench.py --batch 8 --quick --only pre_ 2>&1 | grep -v "^{" | tail -14 | tee -a gpurun_out/run8.log
for b in 8 16; do
  echo "=== bench B=$b (auto gemm mode)" | tee -a gpurun_out/run8.log
  timeout 1200 python bench.py --steps 2 --warmup 1 --batch $b --no-cpu-baseline 2>&1 | tail -1 | python -c "
import sys, json
d = json.loads(sys.stdin.read())
print({k: d[k] for k in ('value','ms_per_step','p50_ttft_ms','ttft_ms_single_clip','stage_ms_instrumented_step')}, d.get('roofline',{}).get('achieved'), d.get('roofline',{}).get('avg_launch_us'))" | tee -a gpurun_out/run8.log
done
