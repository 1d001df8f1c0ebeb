#!/bin/bash
mkdir -p gpurun_out
: > gpurun_out/power_half.log
for flags in "--masked" "--masked --half-grid" "--masked" "--masked --half-grid"; do
  python tools/gpu/power_microbench.py dec_gateup dec_qkv $flags 2>/dev/null | grep -v amdgpu >> gpurun_out/power_half.log
done
cat gpurun_out/power_half.log
