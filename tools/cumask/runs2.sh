cd /root/repo
for cfg in "--overlap 0" "--overlap 1 --front-cus 16 --overlap-steps 12" "--overlap 1 --front-cus 16 --overlap-steps 0" "--overlap 1 --front-cus 16 --overlap-steps 16" "--overlap 1 --front-cus 12 --overlap-steps 14" "--overlap 1 --front-cus 20 --overlap-steps 10"; do
  echo "=== $cfg"
  timeout 900 python bench.py --steps 2 --warmup 1 --no-cpu-baseline --no-instrument $cfg 2>&1 | tail -3 | python tools/cumask/show.py
done
