cd /root/repo
python -m pytest tests/test_gpu_kernels.py tests/test_gpu_prefill_batch.py tests/test_gpu_vit.py -x -q 2>&1 | tail -2
for cfg in "--gemm-tile-order 0" "--gemm-tile-order 1" "--gemm-tile-order 0" "--gemm-tile-order 1"; do
  echo "=== $cfg"
  timeout 900 python bench.py --steps 2 --warmup 1 --no-cpu-baseline --no-instrument $cfg 2>&1 | tail -3 | python tools/cumask/show.py
done
python tools/microbench.py --batch 8 --nseq 8 --only pre_ 2>&1 | grep -E "LDS epilogue"
