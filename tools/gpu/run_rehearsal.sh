#!/bin/bash
# What the driver runs at round end: pytest -m gpu, smoke(), default bench.
mkdir -p gpurun_out
echo "=== pytest -m gpu" | tee gpurun_out/rehearsal.log
timeout 1500 python -m pytest tests/ -x -q -m gpu --no-header -p no:cacheprovider 2>&1 | tail -5 | tee -a gpurun_out/rehearsal.log
echo "=== smoke" | tee -a gpurun_out/rehearsal.log
timeout 600 python -c "import __graft_entry__ as g; g.smoke()" 2>&1 | tail -2 | tee -a gpurun_out/rehearsal.log
echo "=== default bench" | tee -a gpurun_out/rehearsal.log
T0=$(date +%s); timeout 1500 python bench.py 2> gpurun_out/bench_time.txt | tail -1 > gpurun_out/bench_default.json; echo "bench wall seconds: $(( $(date +%s) - T0 ))" | tee -a gpurun_out/rehearsal.log
tail -3 gpurun_out/bench_time.txt | cut -c1-300 | tee -a gpurun_out/rehearsal.log
python - <<'PY' | tee -a gpurun_out/rehearsal.log
import json
d = json.load(open("gpurun_out/bench_default.json"))
for k in ("value", "ms_per_step", "p50_ttft_ms", "ttft_ms_single_clip", "stage_ms_instrumented_step"):
    print(k, d.get(k))
for k in ("roofline", "roofline_decode_gateup", "cpu_baseline"):
    v = d.get(k) or {}
    print(k, {x: v.get(x) for x in ("kernel", "achieved", "frac", "traffic", "algorithmic_bytes_per_launch", "avg_launch_us", "value", "cores", "s_per_caption", "ttft_s")})
PY
