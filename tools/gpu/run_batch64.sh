#!/bin/bash
# GPU: decode batches beyond 32 slots (3-4 MFMA column groups): parity tests, kernel microbench, whole-path bench.
mkdir -p gpurun_out
: > gpurun_out/batch64.log
timeout 900 python -m pytest tests/test_gpu_kernels.py tests/test_gpu_prefill_batch.py -x -q 2>&1 | tail -4 | tee -a gpurun_out/batch64.log
for b in 32 48 64; do
  echo "=== microbench B=$b" | tee -a gpurun_out/batch64.log
  timeout 600 python tools/microbench.py --batch $b --quick --only dec_ 2>&1 | grep -v amdgpu.ids | tee -a gpurun_out/batch64.log
done
for b in 32 48 64; do
  echo "=== bench B=$b" | tee -a gpurun_out/batch64.log
  timeout 1500 python bench.py --steps 2 --warmup 1 --batch $b --no-cpu-baseline 2>gpurun_out/batch64_err_$b.txt | tail -1 | python -c "
import sys, json
d = json.loads(sys.stdin.read())
print({k: d.get(k) for k in ('value','ms_per_step','p50_ttft_ms','stage_ms_instrumented_step')})
print({k: (d.get('roofline') or {}).get(k) for k in ('achieved','frac','avg_launch_us')})" | tee -a gpurun_out/batch64.log
  tail -2 gpurun_out/batch64_err_$b.txt | cut -c1-300 | tee -a gpurun_out/batch64.log
done
