#!/bin/bash
# per-kernel time of one clip's ViT-H + ToMe pass:  tools/gpu/vit_trace.sh [frames] [tag]
REPO=${GRAFT_REPO_ROOT:-$(pwd)}
F=${1:-8}; TAG=${2:-vit$F}
OUT=$REPO/gpurun_out/prof_$TAG
rm -rf "$OUT"; mkdir -p "$OUT"
cd /tmp && export TMPDIR=/tmp
python $REPO/tools/vit_single_clip.py $F 20 > "$OUT/plain.log" 2>&1
timeout 600 rocprofv3 --kernel-trace --stats --output-format csv -d "$OUT/trace" -o trace -- python $REPO/tools/vit_single_clip.py $F 20 > "$OUT/trace.log" 2>&1
python "$REPO/tools/rocprof_summary.py" stats "$OUT/trace" "$OUT/${TAG}_kernel_stats.txt" > /dev/null
find "$OUT" -name "*.csv" -size +2M -delete; find "$OUT" -name "*.db" -delete
grep "ms per pass" "$OUT/plain.log" "$OUT/trace.log"
head -30 "$OUT/${TAG}_kernel_stats.txt"
