#!/bin/bash
mkdir -p gpurun_out
: > gpurun_out/r03_halfgrid.log
for f in "" "--half-grid" "" "--half-grid" "" "--half-grid"; do
  timeout 900 python bench.py $f --steps 2 --warmup 1 --no-cpu-baseline --no-instrument 2>/dev/null | python -c "
import sys,json
d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('[$f]', round(d['value'],3), 'captions/s', round(d['ms_per_step'],1), 'ms/step', d['power']['sclk_mhz_p50'], d['power']['socket_power_w_p50'])" >> gpurun_out/r03_halfgrid.log
done
cat gpurun_out/r03_halfgrid.log
