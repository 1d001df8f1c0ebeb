#!/bin/bash
# rocprofv3 kernel trace of one short step (32 new tokens), grouped by (kernel, grid): in-situ time of every GEMM shape
R=$GRAFT_REPO_ROOT; [ -z "$R" ] && R=$(pwd)
mkdir -p $R/gpurun_out; cd /tmp && export TMPDIR=/tmp
timeout 900 rocprofv3 --kernel-trace --output-format csv -d /tmp/ps -o r1 -- python $R/bench.py --steps 1 --warmup 0 --max_new_tokens 32 --no-cpu-baseline --no-instrument > /tmp/ps.log 2>&1
python $R/tools/rocprof_summary.py shapes /tmp/ps $R/gpurun_out/kernel_shapes.txt | cut -c1-190
