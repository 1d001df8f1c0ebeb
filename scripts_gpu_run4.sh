#!/bin/bash
mkdir -p gpurun_out
./scripts_gpu_tests.sh > /dev/null 2>&1; echo "tests rc=$?" | tee gpurun_out/run4.log; grep -E "passed|failed|error" gpurun_out/tests.log | tee -a gpurun_out/run4.log
for b in 1 16; do
  echo "=== bench B=$b" | tee -a gpurun_out/run4.log
  timeout 1200 python bench.py --steps 2 --warmup 1 --batch $b --no-cpu-baseline 2>&1 | tail -1 | python -c "
import sys, json
d = json.loads(sys.stdin.read())
print({k: d[k] for k in ('value','ms_per_step','p50_ttft_ms','ttft_ms_single_clip','stage_ms_instrumented_step')}, d.get('roofline',{}).get('achieved'), d.get('roofline',{}).get('avg_launch_us'))" | tee -a gpurun_out/run4.log
done
echo "=== bench B=8 eager" | tee -a gpurun_out/run4.log
timeout 1200 python bench.py --steps 2 --warmup 1 --batch 8 --no-cpu-baseline --no-graph --no-instrument 2>&1 | tail -1 | cut -c1-400 | tee -a gpurun_out/run4.log
