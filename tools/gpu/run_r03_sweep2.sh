#!/bin/bash
# round 3, second sweep: finer grid around the default schedule (overlap steps 6-8 per chunk, front end on 14 / 16 / 18 CUs per XCD)
mkdir -p gpurun_out
: > gpurun_out/r03_sched_sweep2.log
run() {
  timeout 700 python bench.py --steps 2 --warmup 1 --no-cpu-baseline --no-instrument "$@" 2>/dev/null | python -c "
import sys,json
d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('$*', '->', round(d['value'],3), 'captions/s', round(d['ms_per_step'],1), 'ms/step, p50 TTFT', round(d['p50_ttft_ms'],1), 'sclk', d.get('power',{}).get('sclk_mhz_p50'))" >> gpurun_out/r03_sched_sweep2.log
}
run
run --overlap-steps 7
run --overlap-steps 8
run --front-cus 14
run --front-cus 18
run --front-cus 14 --overlap-steps 8
run --front-cus 18 --overlap-steps 7
run --overlap-steps 7
run
cat gpurun_out/r03_sched_sweep2.log
