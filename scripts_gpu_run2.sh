#!/bin/bash
mkdir -p gpurun_out
echo "=== llm tests" | tee gpurun_out/run2.log
timeout 900 python -m pytest tests/test_gpu_llm.py -q -m gpu --no-header -p no:cacheprovider 2>&1 | tail -30 | tee -a gpurun_out/run2.log
echo "=== smoke" | tee -a gpurun_out/run2.log
timeout 600 python -c "import __graft_entry__ as g; g.smoke()" 2>&1 | tail -5 | tee -a gpurun_out/run2.log
echo "=== tiny bench" | tee -a gpurun_out/run2.log
timeout 600 python bench.py --tiny --steps 2 --warmup 1 --batch 4 --max_new_tokens 32 2>&1 | tail -5 | tee -a gpurun_out/run2.log
echo "=== real bench B=8" | tee -a gpurun_out/run2.log
timeout 1500 python bench.py --steps 1 --warmup 1 --batch 8 2>&1 | tail -8 | tee -a gpurun_out/run2.log
