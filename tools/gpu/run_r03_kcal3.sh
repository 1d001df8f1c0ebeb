#!/bin/bash
# schedule rule + calibration inside the warm-up: the four configs by name, then the bench-launch tests
mkdir -p gpurun_out
: > gpurun_out/r03_kcal3.log
run() {
  timeout 1500 python bench.py --warmup 1 --no-cpu-baseline "$@" 2>gpurun_out/kcal.err | tail -1 > gpurun_out/kcal_line.json; python -c "
import sys,json
d=json.loads(open('gpurun_out/kcal_line.json').read()); print('$*', '->', round(d['value'],3), 'captions/s', round(d['ms_per_step'],1), 'ms/step, p50 TTFT', round(d['p50_ttft_ms'],1), d.get('overlap_steps_calibration'), d['config']['mode'][:40], (d.get('roofline_step') or {}).get('frac'))" >> gpurun_out/r03_kcal3.log 2>&1 || tail -5 gpurun_out/kcal.err >> gpurun_out/r03_kcal3.log
}
run --steps 2; cp gpurun_out/kcal_line.json gpurun_out/r03_bench_default.json
run --config cfg4 --steps 1; cp gpurun_out/kcal_line.json gpurun_out/r03_bench_cfg4.json
run --config cfg3 --steps 1; cp gpurun_out/kcal_line.json gpurun_out/r03_bench_cfg3.json
run --config cfg5 --steps 1; cp gpurun_out/kcal_line.json gpurun_out/r03_bench_cfg5.json
python -m pytest tests/test_gpu_bench_launch.py -x -q -m gpu 2>&1 | grep -v amdgpu.ids | tail -5 >> gpurun_out/r03_kcal3.log
cat gpurun_out/r03_kcal3.log
