#!/bin/bash
# default config: calibrated overlap steps vs the fixed rule, alternating on one box
mkdir -p gpurun_out
: > gpurun_out/r03_kcal2.log
run() {
  timeout 1500 python bench.py --steps 2 --warmup 1 --no-cpu-baseline --no-instrument "$@" 2>gpurun_out/kcal.err | python -c "
import sys,json
d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('$*', '->', round(d['value'],3), 'captions/s', round(d['ms_per_step'],1), 'ms/step, p50 TTFT', round(d['p50_ttft_ms'],1), d.get('overlap_steps_calibration'), d.get('power',{}).get('sclk_mhz_p50'), d.get('power',{}).get('socket_power_w_p50'))" >> gpurun_out/r03_kcal2.log 2>&1 || tail -5 gpurun_out/kcal.err >> gpurun_out/r03_kcal2.log
}
run --overlap-steps 6
run
run --overlap-steps 6
run
run --overlap-steps 6
cat gpurun_out/r03_kcal2.log
