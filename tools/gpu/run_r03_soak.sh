#!/bin/bash
# the whole GPU suite three times in a row on one box (flaky-test screen), then smoke
mkdir -p gpurun_out
: > gpurun_out/r03_soak.log
for i in 1 2 3; do
  ( time timeout 1500 python -m pytest tests/ -x -q -m gpu --no-header -p no:cacheprovider 2>&1 | grep -v amdgpu.ids | tail -6 ) >> gpurun_out/r03_soak.log 2>&1
done
timeout 300 python -c "import __graft_entry__ as g; g.smoke()" 2>&1 | grep -v amdgpu.ids | tail -2 >> gpurun_out/r03_soak.log
cat gpurun_out/r03_soak.log
