#!/bin/bash
# Device-side AddressSanitizer over the cross-workgroup hand-over kernels.
# Build first (build container or GPU box): python -m aurora_amd.build --device-asan   -> aurora_amd/libaurora_hip_dasan.so
# (host + gfx950:xnack+ code instrumented, shared ASan runtime).  Runs the ToMe hand-over tests and the split-K hand-over tests with
# HSA_XNACK=1 and the runtime preloaded; whatever happens (a report, a clean pass, or the reason the box cannot run an xnack+ sanitized
# code object) lands in gpurun_out/device_asan.log.
mkdir -p gpurun_out
LOG=gpurun_out/device_asan.log
: > $LOG
RT=$(ls /opt/rocm/lib/llvm/lib/clang/*/lib/linux/libclang_rt.asan-x86_64.so | tail -1)
echo "asan runtime: $RT" >> $LOG
echo "instrumented ROCm runtime (/opt/rocm/lib/asan): $(ls -d /opt/rocm/lib/asan 2>/dev/null || echo absent)" >> $LOG
rocminfo 2>/dev/null | grep -iE "xnack|gfx950" | sort | uniq -c >> $LOG
for mode in "HSA_XNACK=1" "HSA_XNACK=0"; do
  echo "=== $mode" >> $LOG
  env $mode LD_PRELOAD=$RT ASAN_OPTIONS=detect_leaks=0:protect_shadow_gap=0:abort_on_error=0 AURORA_HIP_SO=$PWD/aurora_amd/libaurora_hip_dasan.so \
    timeout 900 python -m pytest tests/test_gpu_kernels.py -q -m gpu -x --no-header -p no:cacheprovider -k "tome or hand" 2>&1 | grep -v amdgpu.ids | tail -25 >> $LOG
  echo "--- split-K hand-over (decode_fused_reduce) under the sanitizer" >> $LOG
  env $mode LD_PRELOAD=$RT ASAN_OPTIONS=detect_leaks=0:protect_shadow_gap=0:abort_on_error=0 AURORA_HIP_SO=$PWD/aurora_amd/libaurora_hip_dasan.so \
    timeout 900 python -m pytest tests/test_gpu_skinny_lds.py -q -m gpu -x --no-header -p no:cacheprovider -k "fused" 2>&1 | grep -v amdgpu.ids | tail -25 >> $LOG
done
cat $LOG
