cd /root/repo
python -m pytest tests/test_gpu_skinny_lds.py tests/test_gpu_llm.py tests/test_gpu_prefill_batch.py tests/test_gpu_caption_batch.py -x -q 2>&1 | tail -3
for b in 64 128; do echo "== B=$b"; python tools/microbench.py --batch $b --only dec_ 2>&1 | grep -E "^decode x-through-LDS"; done
echo "== B=128 masked 16, full grid"; python tools/microbench.py --batch 128 --only dec_ --mask-cus 16 2>&1 | grep -E "^decode x-through-LDS"
for cfg in "--no-half-grid" "" "--no-half-grid" ""; do
  echo "=== $cfg"
  timeout 900 python bench.py --steps 2 --warmup 1 --no-cpu-baseline --no-instrument $cfg 2>&1 | tail -3 | python tools/cumask/show.py
done
