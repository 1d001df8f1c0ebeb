#!/bin/bash
mkdir -p gpurun_out
: > gpurun_out/r03_permlane.log
python -m pytest tests/test_gpu_kernels.py tests/test_gpu_llm.py tests/test_gpu_vit.py tests/test_gpu_skinny_lds.py -x -q -m gpu 2>&1 | grep -v amdgpu.ids | tail -5 >> gpurun_out/r03_permlane.log
for i in 1 2; do
  AURORA_HIP_SO=$PWD/aurora_amd/libaurora_hip_old.so python tools/gpu/ab_microbench.py 2>/dev/null | grep -v amdgpu >> gpurun_out/r03_permlane.log
  python tools/gpu/ab_microbench.py 2>/dev/null | grep -v amdgpu >> gpurun_out/r03_permlane.log
done
for so in old new old new; do
  if [ $so = old ]; then export AURORA_HIP_SO=$PWD/aurora_amd/libaurora_hip_old.so; else unset AURORA_HIP_SO; fi
  python bench.py --no-cpu-baseline --no-instrument 2>/dev/null | python -c "
import sys,json
d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('$so', round(d['value'],3), round(d['ms_per_step'],1), d['power']['sclk_mhz_p50'], d['power']['socket_power_w_p50'])" >> gpurun_out/r03_permlane.log
done
cat gpurun_out/r03_permlane.log
