#!/bin/bash
# batched prefill: correctness + bench with prefill groups
mkdir -p gpurun_out
echo "=== tests" | tee gpurun_out/run10.log
timeout 900 python -m pytest tests/test_gpu_prefill_batch.py tests/test_gpu_llm.py -q -m gpu --no-header -p no:cacheprovider 2>&1 | tail -8 | tee -a gpurun_out/run10.log
for cfgs in "8 1" "8 4" "8 8" "16 4" "16 16"; do
  set -- $cfgs
  echo "=== bench B=$1 prefill-group=$2" | tee -a gpurun_out/run10.log
  timeout 1200 python bench.py --steps 2 --warmup 1 --batch $1 --prefill-group $2 --no-cpu-baseline 2>&1 | tail -1 | python -c "
import sys, json
d = json.loads(sys.stdin.read())
print({k: d[k] for k in ('value','ms_per_step','p50_ttft_ms','ttft_ms_single_clip','stage_ms_instrumented_step')})" | tee -a gpurun_out/run10.log
done
