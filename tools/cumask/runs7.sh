cd /root/repo
python -m pytest tests/test_gpu_caption_batch.py -x -q 2>&1 | tail -2
for cfg in "--overlap-steps 5" "--overlap-steps 7" "--overlap-steps 6" "--prefill-group 8 --overlap-steps 12"; do
  echo "=== $cfg"
  timeout 900 python bench.py --steps 2 --warmup 1 --no-cpu-baseline --no-instrument $cfg 2>&1 | tail -3 | python tools/cumask/show.py
done
