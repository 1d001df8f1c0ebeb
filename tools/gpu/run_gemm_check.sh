#!/bin/bash
# GPU: GEMM parity tests + prefill GEMM microbench at the batched shape (8 sequences per pass) + a short whole-path bench.
mkdir -p gpurun_out
: > gpurun_out/gemm.log
timeout 900 python -m pytest tests/test_gpu_kernels.py tests/test_gpu_prefill_batch.py tests/test_gpu_vit.py -x -q 2>&1 | tail -4 | tee -a gpurun_out/gemm.log
echo "=== microbench prefill nseq=8" | tee -a gpurun_out/gemm.log
timeout 900 python tools/microbench.py --batch 8 --nseq 8 --quick --only pre_ 2>&1 | grep -v "amdgpu.ids\|^{" | tee -a gpurun_out/gemm.log
B=${1:-64}
echo "=== bench B=$B" | tee -a gpurun_out/gemm.log
timeout 1500 python bench.py --steps 2 --warmup 1 --batch $B --no-cpu-baseline 2>gpurun_out/gemm_err.txt | tail -1 | python -c "
import sys, json
d = json.loads(sys.stdin.read())
print({k: d.get(k) for k in ('value','ms_per_step','p50_ttft_ms','ttft_ms_single_clip','stage_ms_instrumented_step')})" | tee -a gpurun_out/gemm.log
tail -2 gpurun_out/gemm_err.txt | cut -c1-300 | tee -a gpurun_out/gemm.log
