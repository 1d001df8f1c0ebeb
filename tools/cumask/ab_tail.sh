cd /root/repo
python -m pytest tests/test_gpu_kernels.py tests/test_gpu_prefill_batch.py tests/test_gpu_vit.py tests/test_gpu_llm.py -x -q 2>&1 | tail -2
for cfg in "--gemm-tail-split 0" "--gemm-tail-split 1" "--gemm-tail-split 0" "--gemm-tail-split 1"; do
  echo "=== $cfg"
  timeout 900 python bench.py --steps 2 --warmup 1 --no-cpu-baseline --no-instrument $cfg 2>&1 | tail -3 | python tools/cumask/show.py
done
