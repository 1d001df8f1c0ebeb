#!/bin/bash
# round-5 baseline: the B = 1 regime nobody has measured (VERDICT r4 missing #2)
mkdir -p gpurun_out/r05
python bench.py --batch 1 --steps 3 --warmup 1 --no-cpu-baseline > gpurun_out/r05/bench_b1.json 2> gpurun_out/r05/bench_b1.err
tail -c 3000 gpurun_out/r05/bench_b1.json
BENCH_FLAGS="--batch 1" bash tools/gpu/step_timeline.sh > gpurun_out/r05/timeline_b1.log 2>&1
cp gpurun_out/prof_cfg4/cfg4_step_timeline.txt gpurun_out/r05/b1_step_timeline.txt
cat gpurun_out/r05/b1_step_timeline.txt
timeout 600 python tools/microbench.py --batch 1 --quick --only dec 2>&1 | grep -v amdgpu.ids > gpurun_out/r05/micro_b1.log
cat gpurun_out/r05/micro_b1.log
