#!/bin/bash
# round-5 validation: full GPU suite (parity record) + the default bench line + the one-slot line
mkdir -p gpurun_out/r05
rm -f gpurun_out/tests.log gpurun_out/parity_observed.json
bash tools/gpu/lab.sh tests > gpurun_out/r05/tests_summary.log 2>&1
tail -20 gpurun_out/r05/tests_summary.log
python bench.py --steps 2 --warmup 1 --no-cpu-baseline > gpurun_out/r05/bench_default.json 2> gpurun_out/r05/bench_default.err
tail -5 gpurun_out/r05/bench_default.err
python - <<'PY'
import json
d = json.loads(open("gpurun_out/r05/bench_default.json").read().strip().splitlines()[-1])
keep = {k: d.get(k) for k in ("value", "ms_per_step", "p50_ttft_ms", "ttft_ms_single_clip", "host", "decode_step", "frontier", "single_stream", "power")}
keep["roofline_frac"] = d["roofline"]["frac"]; keep["tome"] = d.get("roofline_tome"); keep["ttft_stage_ms"] = d.get("ttft_stage_ms")
keep["vit"] = d["roofline_vit"]["frac"]; keep["step"] = d["roofline_step"]["frac"]; keep["cal"] = d.get("overlap_steps_calibration")
print(json.dumps(keep, indent=1))
PY
python bench.py --batch 1 --steps 3 --warmup 1 --no-cpu-baseline > gpurun_out/r05/bench_b1.json 2> gpurun_out/r05/bench_b1.err
python - <<'PY'
import json
d = json.loads(open("gpurun_out/r05/bench_b1.json").read().strip().splitlines()[-1])
print("B=1:", d["value"], d["ms_per_step"], d.get("decode_step"), d.get("ttft_stage_ms", {}).get("single_clip"))
PY
