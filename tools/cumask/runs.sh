set -x
cd /root/repo
for cfg in "" "--front-cus 16" "--front-cus 16 --decode-cus 16" "--front-cus 12 --decode-cus 20" "--front-cus 20 --decode-cus 12" "--front-cus 24"; do
  echo "=== $cfg"
  timeout 600 python bench.py --batch 64 --pipeline --steps 2 --warmup 1 --no-cpu-baseline $cfg 2>&1 | tail -1 | python -c "
import sys,json
for l in sys.stdin:
    try:
        d=json.loads(l); print('value',d['value'],'ms_per_step',d['ms_per_step'],'ttft',d.get('ttft_ms_p50'))
    except Exception as e: print('ERR',l[:300])
"
done
