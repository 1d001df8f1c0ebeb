#!/bin/bash
# rocprofv3 passes for the bench command; summaries land in gpurun_out/prof_*.txt
R=$GRAFT_REPO_ROOT; [ -z "$R" ] && R=$(pwd)
mkdir -p $R/gpurun_out
cd /tmp && export TMPDIR=/tmp
echo "=== kernel trace" | tee -a $R/gpurun_out/run3.log
timeout 1500 rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/prof_stats -o r1 -- python $R/bench.py --steps 1 --warmup 0 --batch 8 --no-cpu-baseline --no-instrument > /tmp/prof_stats.log 2>&1
grep -v "^W2026" /tmp/prof_stats.log | tail -12 | cut -c1-400 | tee -a $R/gpurun_out/run3.log
python $R/tools/rocprof_summary.py stats /tmp/prof_stats $R/gpurun_out/prof_kernel_stats.txt | head -36 | tee -a $R/gpurun_out/run3.log
for c in FETCH_SIZE WRITE_SIZE; do
  echo "=== pmc $c" | tee -a $R/gpurun_out/run3.log
  timeout 1200 rocprofv3 --pmc $c --output-format csv -d /tmp/prof_$c -o r1 -- python $R/bench.py --steps 1 --warmup 0 --batch 8 --max_new_tokens 6 --no-cpu-baseline --no-instrument --no-graph > /tmp/prof_$c.log 2>&1
  grep -v "^W2026" /tmp/prof_$c.log | tail -6 | cut -c1-300 | tee -a $R/gpurun_out/run3.log
  python $R/tools/rocprof_summary.py pmc /tmp/prof_$c $R/gpurun_out/prof_pmc_$c.txt | head -45 | tee -a $R/gpurun_out/run3.log
done
ls -la /tmp/prof_stats /tmp/prof_stats/* | head -20 >> $R/gpurun_out/run3.log
