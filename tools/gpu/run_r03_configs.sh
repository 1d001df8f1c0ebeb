#!/bin/bash
# round 3: every BASELINE config on the current tree, by name -> profiles/r03_bench_<cfg>.json
mkdir -p gpurun_out
for c in cfg3 cfg5 cfg4; do
  timeout 2400 python bench.py --config $c --steps 1 --warmup 1 --no-cpu-baseline > gpurun_out/r03_bench_$c.json 2> gpurun_out/r03_bench_$c.err
  tail -c 300 gpurun_out/r03_bench_$c.err
  python - <<PY
import json
try:
    d = json.loads(open("gpurun_out/r03_bench_$c.json").read().strip().splitlines()[-1])
    print("$c", {k: d.get(k) for k in ("value", "ms_per_step", "p50_ttft_ms", "p50_ttft_host_ms")}, d["config"]["workload"], d["config"]["clips_per_gpu_per_step"],
          {k: (d.get("roofline") or {}).get(k) for k in ("frac", "avg_launch_us")}, (d.get("roofline_step") or {}).get("frac"))
except Exception as e:
    print("$c failed", e)
PY
done
