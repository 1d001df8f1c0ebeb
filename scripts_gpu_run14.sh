#!/bin/bash
# norm-free decode step (folded RMSNorm weights, int64 sum(x^2) carried by the residual epilogues)
mkdir -p gpurun_out
echo "=== tests" | tee gpurun_out/run14.log
for f in tests/test_gpu_kernels.py tests/test_gpu_llm.py tests/test_gpu_prefill_batch.py; do
  timeout 900 python -m pytest $f -q -m gpu --no-header -p no:cacheprovider 2>&1 | tail -14 | tee -a gpurun_out/run14.log
done
timeout 300 python -c "import __graft_entry__ as g; g.smoke()" 2>&1 | tail -2 | tee -a gpurun_out/run14.log
echo "=== microbench B=32" | tee -a gpurun_out/run14.log
timeout 900 python tools/microbench.py --batch 32 --quick --only dec_ 2>&1 | grep -v "^{" | tail -8 | tee -a gpurun_out/run14.log
for cfgs in "8 4" "16 4" "32 8"; do
  set -- $cfgs
  echo "=== bench B=$1 prefill-group=$2" | tee -a gpurun_out/run14.log
  timeout 1200 python bench.py --steps 2 --warmup 1 --batch $1 --prefill-group $2 --no-cpu-baseline 2>&1 | tail -1 | python -c "
import sys, json
d = json.loads(sys.stdin.read())
print({k: d[k] for k in ('value','ms_per_step','p50_ttft_ms','ttft_ms_single_clip','stage_ms_instrumented_step')}, d['roofline']['achieved'], d['roofline']['avg_launch_us'])" | tee -a gpurun_out/run14.log
done
