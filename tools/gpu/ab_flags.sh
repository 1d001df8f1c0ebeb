#!/bin/bash
# alternating bench.py runs over a list of flag strings; prints the stage figures that matter for front-end A/Bs
#   tools/gpu/ab_flags.sh <pairs> "<flags A>" "<flags B>" ...
n=$1; shift
for i in $(seq $n); do
  for f in "$@"; do
    timeout 900 python bench.py --steps 2 --no-cpu-baseline $f > gpurun_out/abf.json 2> gpurun_out/abf.err
    python - "$f" <<'PY'
import json, sys
d = json.loads(open('gpurun_out/abf.json').read().strip().splitlines()[-1])
pp = d['roofline_prefill_gemm']['per_projection_tflops']
print(f"[{sys.argv[1]:18s}] captions/s {d['value']:.3f}  single clip {d['ttft_ms_single_clip']:.2f} ms  vit {d['roofline_vit']['achieved']:.1f} TF/s  prefill gemm {d['roofline_prefill_gemm']['achieved']:.1f}"
      f" (qkv {pp['pre_qkv']:.0f} o {pp['pre_o']:.0f} gu {pp['pre_gateup']:.0f} down {pp['pre_down']:.0f})  single-clip stages "
      + " ".join(f"{k} {v:.2f}" for k, v in d['ttft_stage_ms']['single_clip'].items() if isinstance(v, float)), flush=True)
PY
  done
done
