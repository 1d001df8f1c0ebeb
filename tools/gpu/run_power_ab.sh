#!/bin/bash
mkdir -p gpurun_out
: > gpurun_out/power_ab.log
for so in old lab old lab; do
  AURORA_HIP_SO=$PWD/aurora_amd/libaurora_hip_$so.so python tools/gpu/power_microbench.py "$@" 2>/dev/null | grep -v amdgpu >> gpurun_out/power_ab.log
done
cat gpurun_out/power_ab.log
