#!/bin/bash
# round 3 final: smoke, the whole GPU suite (one pytest process, as the driver runs it), the default bench with the CPU baseline
mkdir -p gpurun_out
timeout 300 python -c "import __graft_entry__ as g; g.build(); g.smoke()" 2>&1 | grep -v amdgpu.ids | tail -3 > gpurun_out/r03_final_tests.log
( time timeout 2400 python -m pytest tests/ -x -q -m gpu --no-header -p no:cacheprovider 2>&1 | grep -v amdgpu.ids | tail -8 ) >> gpurun_out/r03_final_tests.log 2>&1
timeout 2400 python bench.py > gpurun_out/r03_bench_default.json 2> gpurun_out/r03_bench_default.err
cat gpurun_out/r03_final_tests.log
python - <<'PY'
import json
d = json.loads(open("gpurun_out/r03_bench_default.json").read().strip().splitlines()[-1])
print({k: d.get(k) for k in ("value", "ms_per_step", "p50_ttft_ms", "p50_ttft_host_ms", "power")})
print(d.get("roofline"))
print(d.get("cpu_baseline", {}).get("value"), d.get("cpu_baseline", {}).get("cores"))
PY
