#!/bin/bash
# rocprofv3 --kernel-trace of the DEFAULT (overlapped) schedule, on the GPU box:  gpurun -- 'bash tools/profile_overlap.sh r02'
TAG=${1:-r02}
REPO=${GRAFT_REPO_ROOT:-$(pwd)}
OUT=$REPO/gpurun_out/prof_overlap_$TAG
rm -rf "$OUT"; mkdir -p "$OUT"
cd /tmp && export TMPDIR=/tmp
for attempt in 1 2 3; do
  rm -rf "$OUT/trace"
  timeout 1200 rocprofv3 --kernel-trace --output-format csv -d "$OUT/trace" -o ovl -- python $REPO/bench.py --no-cpu-baseline --no-power --no-instrument --steps 1 --warmup 1 --sync-chunks > "$OUT/trace.log" 2>&1 && break
done
python $REPO/tools/overlap_trace_summary.py "$OUT/trace" "$OUT/${TAG}_overlap_trace_summary.txt"
find "$OUT" -name "*.csv" -size +2M -delete
find "$OUT" -name "*.db" -delete
cat "$OUT/${TAG}_overlap_trace_summary.txt"
grep -o '"value": [0-9.]*' "$OUT/trace.log" | head -2
