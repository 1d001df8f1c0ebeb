#!/bin/bash
# BASELINE.json configs[2] (16 frames, ratio 0.2, 512 tokens) and configs[4] (ratio 0.8, 2048 tokens, long KV) for the record
mkdir -p gpurun_out
: > gpurun_out/other_configs.log
run() {
  echo "=== $*" | tee -a gpurun_out/other_configs.log
  timeout 1500 python bench.py --steps 1 --warmup 1 --no-cpu-baseline "$@" 2>gpurun_out/other_err.txt | tail -1 | python -c "
import sys, json
d = json.loads(sys.stdin.read())
print({k: d.get(k) for k in ('value','ms_per_step','p50_ttft_ms','ttft_ms_single_clip','stage_ms_instrumented_step')})
print({k: d['config'].get(k) for k in ('clips_per_gpu_per_step','frames','r_per_layer','visual_tokens_per_clip','prefill_len','max_new_tokens')})
print({k: (d.get('roofline') or {}).get(k) for k in ('kernel','achieved','frac','avg_launch_us')})" | tee -a gpurun_out/other_configs.log
  tail -2 gpurun_out/other_err.txt | cut -c1-300 | tee -a gpurun_out/other_configs.log
}
run --num_frm 16 --token_kept_ratio 0.2 --max_new_tokens 512 --batch 64
run --num_frm 8 --token_kept_ratio 0.8 --max_new_tokens 2048 --batch 32
