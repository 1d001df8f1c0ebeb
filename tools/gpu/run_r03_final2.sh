#!/bin/bash
# half grid on the masked steps as the default: stream / bench tests, then the bench lines of the configs that overlap
mkdir -p gpurun_out
: > gpurun_out/r03_final2.log
timeout 1500 python -m pytest tests/test_gpu_caption_batch.py tests/test_gpu_bench_launch.py tests/test_gpu_skinny_lds.py tests/test_asan_host.py -x -q -m gpu 2>&1 | grep -v amdgpu.ids | tail -4 >> gpurun_out/r03_final2.log
timeout 2400 python bench.py > gpurun_out/r03_bench_default.json 2> gpurun_out/r03_bench_default.err
for c in cfg3 cfg5; do
  timeout 2400 python bench.py --config $c --steps 1 --warmup 1 --no-cpu-baseline > gpurun_out/r03_bench_$c.json 2> gpurun_out/r03_bench_$c.err
done
python - >> gpurun_out/r03_final2.log <<'PY'
import json
for c in ("default", "cfg3", "cfg5"):
    d = json.loads(open("gpurun_out/r03_bench_%s.json" % c).read().strip().splitlines()[-1])
    print(c, round(d["value"], 3), "captions/s", round(d["ms_per_step"], 1), "ms/step, p50 TTFT", round(d["p50_ttft_ms"], 1), d.get("overlap_steps_calibration"),
          (d.get("roofline") or {}).get("frac"), (d.get("roofline_step") or {}).get("frac"), d.get("power"))
PY
cat gpurun_out/r03_final2.log
