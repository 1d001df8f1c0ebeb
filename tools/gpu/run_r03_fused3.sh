#!/bin/bash
mkdir -p gpurun_out
: > gpurun_out/r03_fused3.log
timeout 600 python -m pytest tests/test_gpu_skinny_lds.py -x -q -m gpu 2>&1 | grep -v amdgpu.ids | tail -3 >> gpurun_out/r03_fused3.log
for i in 1 2 3; do
  timeout 700 python bench.py --fused-reduce 1 --steps 2 --warmup 1 --no-cpu-baseline --no-instrument > gpurun_out/f3.json 2> gpurun_out/f3.err
  echo "run $i rc $?" >> gpurun_out/r03_fused3.log
  python -c "
import json
d=json.loads(open('gpurun_out/f3.json').read().strip().splitlines()[-1]); print(round(d['value'],3), 'captions/s', d['ids_checksum_rank0'])" >> gpurun_out/r03_fused3.log 2>&1 || grep -v amdgpu gpurun_out/f3.err | tail -8 >> gpurun_out/r03_fused3.log
done
cat gpurun_out/r03_fused3.log
