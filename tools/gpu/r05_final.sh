#!/bin/bash
# round-5 record runs: the default line exactly as the driver launches it (timed), the other BASELINE configs, the one-slot line
mkdir -p gpurun_out/r05
t0=$(date +%s)
python bench.py --gpus 1 --steps 20 --warmup 3 > gpurun_out/r05/final_default.json 2> gpurun_out/r05/final_default.err
echo "default (steps 20, warmup 3): $(( $(date +%s) - t0 )) s wall"
t0=$(date +%s)
python bench.py > gpurun_out/r05/final_default_noflags.json 2> gpurun_out/r05/final_default_noflags.err
echo "default (no flags): $(( $(date +%s) - t0 )) s wall"
for c in cfg3 cfg4 cfg5; do
  python bench.py --config $c --steps 2 --warmup 1 --no-cpu-baseline > gpurun_out/r05/final_$c.json 2> gpurun_out/r05/final_$c.err
done
python bench.py --batch 1 --steps 3 --warmup 1 --no-cpu-baseline > gpurun_out/r05/final_b1.json 2> gpurun_out/r05/final_b1.err
python - <<'PY'
import json
for n in ("default", "default_noflags", "cfg3", "cfg4", "cfg5", "b1"):
    try:
        d = json.loads(open(f"gpurun_out/r05/final_{n}.json").read().strip().splitlines()[-1])
        print(n, round(d["value"], 3), "captions/s", round(d["ms_per_step"], 1), "ms/step, p50 TTFT", d.get("p50_ttft_ms"), "roofline", round(d["roofline"]["frac"], 3),
              "step", (d.get("roofline_step") or {}).get("frac"), "decode_step", (d.get("decode_step") or {}).get("frac"), "tome", (d.get("roofline_tome") or {}).get("frac"),
              "single", (d.get("single_stream") or {}).get("frac"), "frontier", [(p["prefill_group"], round(p["captions_per_s"], 2), round(p["p50_ttft_ms"] or 0)) for p in d.get("frontier", [])],
              "cpu", (d.get("cpu_baseline") or {}).get("value"))
    except Exception as e:
        print(n, "FAILED", e)
PY
