#!/bin/bash
# rocprofv3 passes for the DEFAULT bench command (batch 64, prefill-group 8); summaries land in gpurun_out/.
# --decode-chunk 64: rocprofv3 --kernel-trace segfaults when > ~150 hipGraph launches are queued ahead of the GPU.
R=$GRAFT_REPO_ROOT; [ -z "$R" ] && R=$(pwd)
B=${1:-64}
mkdir -p $R/gpurun_out
cd /tmp && export TMPDIR=/tmp
echo "=== kernel trace (one full step, batch $B)" | tee $R/gpurun_out/prof.log
timeout 1500 rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/prof_stats -o r1 -- python $R/bench.py --steps 1 --warmup 0 --batch $B --no-cpu-baseline --no-instrument --decode-chunk 64 > /tmp/prof_stats.log 2>&1
grep -v "^W2026\|^E2026" /tmp/prof_stats.log | tail -3 | cut -c1-300 | tee -a $R/gpurun_out/prof.log
python $R/tools/rocprof_summary.py stats /tmp/prof_stats $R/gpurun_out/prof_kernel_stats.txt | head -34 | cut -c1-175 | tee -a $R/gpurun_out/prof.log
for c in FETCH_SIZE WRITE_SIZE; do
  echo "=== pmc $c" | tee -a $R/gpurun_out/prof.log
  timeout 1200 rocprofv3 --pmc $c --output-format csv -d /tmp/prof_$c -o r1 -- python $R/bench.py --steps 1 --warmup 0 --batch $B --max_new_tokens 6 --no-cpu-baseline --no-instrument --no-graph > /tmp/prof_$c.log 2>&1
  grep -v "^W2026\|^E2026" /tmp/prof_$c.log | tail -2 | cut -c1-200 | tee -a $R/gpurun_out/prof.log
  python $R/tools/rocprof_summary.py pmc /tmp/prof_$c $R/gpurun_out/prof_pmc_$c.txt > /dev/null
done
python $R/tools/pmc_to_json.py /tmp/prof_FETCH_SIZE /tmp/prof_WRITE_SIZE $R/gpurun_out/pmc_traffic.json $B 2145 | tee -a $R/gpurun_out/prof.log
