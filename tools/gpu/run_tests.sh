#!/bin/bash
# Run each GPU test file in its own process (a GPU fault in one file does not hide the others).
mkdir -p gpurun_out
rocminfo | grep -E "gfx|Compute Unit" | head -4 > gpurun_out/gpu.txt 2>&1
rc=0
for f in tests/test_gpu_*.py; do
  echo "=== $f" | tee -a gpurun_out/tests.log
  timeout 900 python -m pytest $f -q -m gpu -x --no-header -p no:cacheprovider 2>&1 | tail -40 | tee -a gpurun_out/tests.log
  [ ${PIPESTATUS[0]} -ne 0 ] && rc=1
done
exit $rc
