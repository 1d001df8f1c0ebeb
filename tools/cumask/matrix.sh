# other operating points of the default (overlapped) schedule: every run checks its ids against a batch-mode step inside bench.py
cd /root/repo
for cfg in "--batch 64" "--batch 96" "--batch 128 --max_new_tokens 128" "--batch 64 --num_frm 16" "--batch 128 --token_kept_ratio 0.5" "--batch 120 --prefill-group 8"; do
  echo "=== $cfg"
  timeout 900 python bench.py --steps 1 --warmup 1 --no-cpu-baseline --no-instrument $cfg 2>&1 | tail -3 | python tools/cumask/show.py
done
