#!/bin/bash
# throughput against the number of KV slots in the default (overlapped) schedule
mkdir -p gpurun_out
: > gpurun_out/r03_slots.log
for b in 64 96 128 112; do
  timeout 900 python bench.py --batch $b --steps 2 --warmup 1 --no-cpu-baseline --no-instrument 2>/dev/null | python -c "
import sys,json
d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('$b slots ->', round(d['value'],3), 'captions/s', round(d['ms_per_step'],1), 'ms/step, p50 TTFT', round(d['p50_ttft_ms'],1), d.get('overlap_steps_calibration'), d['power']['sclk_mhz_p50'], d['power']['socket_power_w_p50'])" >> gpurun_out/r03_slots.log
done
cat gpurun_out/r03_slots.log
