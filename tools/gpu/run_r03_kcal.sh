#!/bin/bash
# calibrated overlap steps: cfg4 / cfg3 / cfg5 / default with the calibration, and with the old fixed 0.8-chunk rule for comparison
mkdir -p gpurun_out
: > gpurun_out/r03_kcal.log
run() {
  timeout 1500 python bench.py --steps 1 --warmup 1 --no-cpu-baseline --no-instrument "$@" 2>gpurun_out/kcal.err | python -c "
import sys,json
d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('$*', '->', round(d['value'],3), 'captions/s', round(d['ms_per_step'],1), 'ms/step, p50 TTFT', round(d['p50_ttft_ms'],1), d.get('overlap_steps_calibration'), 'seq', round(d['sequential_schedule']['captions_per_s'],3), 'batch', round(d['batch_mode']['captions_per_s'],3))" >> gpurun_out/r03_kcal.log 2>&1 || tail -5 gpurun_out/kcal.err >> gpurun_out/r03_kcal.log
}
run --config cfg4
run --config cfg4 --overlap-steps 102
run --config cfg3
run --config cfg3 --overlap-steps 17
run --config cfg5
run --steps 2
run --steps 2 --overlap-steps 6
cat gpurun_out/r03_kcal.log
