cd /root/repo
for cfg in "--overlap 1 --prefill-group 4 --overlap-steps 5" "--overlap 1 --prefill-group 4 --overlap-steps 7" "--overlap 1 --prefill-group 4 --overlap-steps 8" "--overlap 1 --prefill-group 2 --overlap-steps 3" "--overlap 1 --prefill-group 2 --overlap-steps 4" "--overlap 1 --prefill-group 4 --overlap-steps 6 --front-cus 12"; do
  echo "=== $cfg"
  timeout 900 python bench.py --steps 2 --warmup 1 --no-cpu-baseline --no-instrument $cfg 2>&1 | tail -3 | python tools/cumask/show.py
done
