#!/bin/bash
mkdir -p gpurun_out/r05
summ() { python - "$1" "$2" <<'PY'
import json, sys
try:
    d = json.loads(open(sys.argv[1]).read().strip().splitlines()[-1])
    h = d.get("host") or {}
    print(f"{sys.argv[2]}: {d['value']:.3f} captions/s, {d['ms_per_step']:.1f} ms/step, p50 TTFT {d.get('p50_ttft_ms')}, step frac {(d.get('roofline_step') or {}).get('frac')}, "
          f"decode_step {(d.get('decode_step') or {}).get('ms_per_step')} ms frac {(d.get('decode_step') or {}).get('frac')}, enqueue wall {h.get('enqueue_s_per_cycle')} cpu {h.get('enqueue_thread_cpu_s_per_cycle')}")
except Exception as e:
    print(f"{sys.argv[2]}: FAILED ({e})")
PY
}
python bench.py --config cfg4 --steps 3 --warmup 1 --no-cpu-baseline > gpurun_out/r05/bench_cfg4.json 2> gpurun_out/r05/bench_cfg4.err; summ gpurun_out/r05/bench_cfg4.json cfg4
python bench.py --prefill-group 2 --steps 1 --warmup 1 --no-cpu-baseline --no-instrument --no-single-stream > gpurun_out/r05/bench_g2.json 2> gpurun_out/r05/bench_g2.err; summ gpurun_out/r05/bench_g2.json "groups of 2"
python bench.py --steps 1 --warmup 1 --no-cpu-baseline --no-instrument --no-single-stream > gpurun_out/r05/bench_g4.json 2> gpurun_out/r05/bench_g4.err; summ gpurun_out/r05/bench_g4.json "groups of 4"
python bench.py --batch 1 --steps 2 --warmup 1 --no-cpu-baseline --no-instrument --attn-fused-combine 1 > gpurun_out/r05/bench_b1_fc.json 2> gpurun_out/r05/bench_b1_fc.err; summ gpurun_out/r05/bench_b1_fc.json "B=1 fused combine"
python bench.py --batch 1 --steps 2 --warmup 1 --no-cpu-baseline --no-instrument > gpurun_out/r05/bench_b1_nf.json 2> gpurun_out/r05/bench_b1_nf.err; summ gpurun_out/r05/bench_b1_nf.json "B=1 separate combine"
python -m pytest tests/test_gpu_bench_launch.py -q -m gpu -x --no-header -p no:cacheprovider -k "full_rate or free_run" -s 2>&1 | tail -8
python -m pytest tests/test_gpu_configs.py -q -m gpu -x --no-header -p no:cacheprovider -k "free_run" 2>&1 | tail -3
