#!/bin/bash
# pruned last prefill layer: bitwise test, LLM suite, A/B of the default bench
mkdir -p gpurun_out
python -m pytest tests/test_gpu_llm.py -m gpu -x -q 2>&1 | tail -15 > gpurun_out/prune_tests.log
for p in 1 0 1 0; do
  python bench.py --prune-last $p --no-cpu-baseline 2>/dev/null | tail -1 >> gpurun_out/prune_ab.jsonl
done
python - <<'PY' >> gpurun_out/prune_tests.log
import json
for l in open("gpurun_out/prune_ab.jsonl"):
    j = json.loads(l)
    print(j["value"], j["ms_per_step"], j.get("power"))
PY
cat gpurun_out/prune_tests.log
