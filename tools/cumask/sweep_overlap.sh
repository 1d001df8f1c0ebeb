# the sweeps behind profiles/r02_overlap_sweep.log (front-end CUs per XCD, masked decode steps per chunk, group size)
cd /root/repo
for cfg in "--overlap 0" "--overlap 1 --front-cus 16 --overlap-steps 12" "--overlap 1 --front-cus 16 --overlap-steps 0" "--overlap 1 --front-cus 16 --overlap-steps 16" "--overlap 1 --front-cus 12 --overlap-steps 14" "--overlap 1 --front-cus 20 --overlap-steps 10"; do
  echo "=== $cfg"
  timeout 900 python bench.py --steps 2 --warmup 1 --no-cpu-baseline --no-instrument $cfg 2>&1 | tail -3 | python tools/cumask/show.py
done
cd /root/repo
for cfg in "--overlap 1 --front-cus 16 --overlap-steps 13" "--overlap 1 --front-cus 16 --overlap-steps 11" "--overlap 1 --front-cus 14 --overlap-steps 13" "--overlap 1 --front-cus 18 --overlap-steps 12" "--overlap 1 --prefill-group 16 --front-cus 16 --overlap-steps 24" "--overlap 1 --prefill-group 4 --front-cus 16 --overlap-steps 6"; do
  echo "=== $cfg"
  timeout 900 python bench.py --steps 2 --warmup 1 --no-cpu-baseline --no-instrument $cfg 2>&1 | tail -3 | python tools/cumask/show.py
done
cd /root/repo
for cfg in "--overlap 1 --prefill-group 4 --overlap-steps 5" "--overlap 1 --prefill-group 4 --overlap-steps 7" "--overlap 1 --prefill-group 4 --overlap-steps 8" "--overlap 1 --prefill-group 2 --overlap-steps 3" "--overlap 1 --prefill-group 2 --overlap-steps 4" "--overlap 1 --prefill-group 4 --overlap-steps 6 --front-cus 12"; do
  echo "=== $cfg"
  timeout 900 python bench.py --steps 2 --warmup 1 --no-cpu-baseline --no-instrument $cfg 2>&1 | tail -3 | python tools/cumask/show.py
done
