#!/bin/bash
# A/B of two builds on one box: aurora_amd/libaurora_hip_old.so against the current one (tools/gpu/ab_microbench.py <kernels>)
mkdir -p gpurun_out
: > gpurun_out/ab.log
for i in 1 2; do
  AURORA_HIP_SO=$PWD/aurora_amd/libaurora_hip_old.so python tools/gpu/ab_microbench.py "$@" 2>/dev/null | grep -v amdgpu >> gpurun_out/ab.log
  python tools/gpu/ab_microbench.py "$@" 2>/dev/null | grep -v amdgpu >> gpurun_out/ab.log
done
cat gpurun_out/ab.log
