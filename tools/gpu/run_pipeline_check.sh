#!/bin/bash
mkdir -p gpurun_out
: > gpurun_out/pipe.log
for mode in "" "--pipeline" "--pipeline --gemm-mode 0"; do
  for b in 16 32; do
  echo "=== bench B=$b $mode" | tee -a gpurun_out/pipe.log
  timeout 1200 python bench.py --steps 3 --warmup 1 --batch $b $mode --no-cpu-baseline --no-instrument 2>&1 | tail -1 | python -c "
import sys, json
d = json.loads(sys.stdin.read())
print({k: d.get(k) for k in ('value','ms_per_step','p50_ttft_ms')})" | tee -a gpurun_out/pipe.log
  done
done
