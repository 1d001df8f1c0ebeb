#!/bin/bash
# kernel trace of one cfg4 (8-slot) bench step -> per-kernel durations and gaps of its decode steps (tools/step_timeline.py)
REPO=${GRAFT_REPO_ROOT:-$(pwd)}
OUT=$REPO/gpurun_out/prof_cfg4
rm -rf "$OUT"; mkdir -p "$OUT"
cd /tmp && export TMPDIR=/tmp
for attempt in 1 2 3; do
  rm -rf "$OUT/trace"
  timeout 900 rocprofv3 --kernel-trace --output-format csv -d "$OUT/trace" -o trace -- python $REPO/bench.py --config cfg4 --no-cpu-baseline --no-power --no-instrument --steps 1 --warmup 0 --decode-chunk 64 ${BENCH_FLAGS} > "$OUT/trace.log" 2>&1 && break
done
python $REPO/tools/step_timeline.py "$OUT/trace" "$OUT/cfg4_step_timeline.txt"
find "$OUT" -name "*.csv" -size +2M -delete; find "$OUT" -name "*.db" -delete
