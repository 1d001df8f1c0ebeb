#!/bin/bash
mkdir -p gpurun_out
echo "=== llm tests (fused norm + pipelined attention are now defaults)" | tee gpurun_out/run5.log
timeout 900 python -m pytest tests/test_gpu_llm.py tests/test_gpu_kernels.py -q -m gpu --no-header -p no:cacheprovider -x 2>&1 | tail -8 | tee -a gpurun_out/run5.log
for b in 8 16; do
  echo "=== microbench B=$b" | tee -a gpurun_out/run5.log
  timeout 900 python tools/microbench.py --batch $b 2>&1 | grep -v "^{" | tail -40 | tee -a gpurun_out/run5.log
done
