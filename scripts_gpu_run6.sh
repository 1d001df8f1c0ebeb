#!/bin/bash
mkdir -p gpurun_out
echo "=== tests" | tee gpurun_out/run6.log
timeout 900 python -m pytest tests/test_gpu_llm.py tests/test_gpu_kernels.py -q -m gpu --no-header -p no:cacheprovider -x 2>&1 | tail -4 | tee -a gpurun_out/run6.log
echo "=== microbench B=8 quick" | tee -a gpurun_out/run6.log
timeout 900 python tools/microbench.py --batch 8 --quick 2>&1 | grep -v "^{" | tail -22 | tee -a gpurun_out/run6.log
for b in 8 16; do
  echo "=== bench B=$b" | tee -a gpurun_out/run6.log
  timeout 1200 python bench.py --steps 2 --warmup 1 --batch $b --no-cpu-baseline 2>&1 | tail -1 | python -c "
import sys, json
d = json.loads(sys.stdin.read())
print({k: d[k] for k in ('value','ms_per_step','p50_ttft_ms','ttft_ms_single_clip','stage_ms_instrumented_step')}, d.get('roofline',{}).get('achieved'), d.get('roofline',{}).get('avg_launch_us'))" | tee -a gpurun_out/run6.log
done
