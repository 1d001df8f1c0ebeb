cd /root/repo
for cfg in "--overlap 1 --front-cus 16 --overlap-steps 13" "--overlap 1 --front-cus 16 --overlap-steps 11" "--overlap 1 --front-cus 14 --overlap-steps 13" "--overlap 1 --front-cus 18 --overlap-steps 12" "--overlap 1 --prefill-group 16 --front-cus 16 --overlap-steps 24" "--overlap 1 --prefill-group 4 --front-cus 16 --overlap-steps 6"; do
  echo "=== $cfg"
  timeout 900 python bench.py --steps 2 --warmup 1 --no-cpu-baseline --no-instrument $cfg 2>&1 | tail -3 | python tools/cumask/show.py
done
