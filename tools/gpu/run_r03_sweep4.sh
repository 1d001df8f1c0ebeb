#!/bin/bash
mkdir -p gpurun_out
: > gpurun_out/r03_sched_sweep4.log
run() {
  timeout 700 python bench.py --steps 2 --warmup 1 --no-cpu-baseline --no-instrument "$@" 2>/dev/null | python -c "
import sys,json
d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('$*', '->', round(d['value'],3), 'captions/s', round(d['ms_per_step'],1), 'ms/step, p50 TTFT', round(d['p50_ttft_ms'],1), d['power']['sclk_mhz_p50'], d['power']['socket_power_w_p50'])" >> gpurun_out/r03_sched_sweep4.log
}
run --overlap-steps 6
run --overlap-steps 7
run --overlap-steps 6
run --overlap-steps 7
run --overlap-steps 6
run --overlap-steps 7
cat gpurun_out/r03_sched_sweep4.log
