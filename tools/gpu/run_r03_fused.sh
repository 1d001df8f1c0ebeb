#!/bin/bash
# fused split-K reduce: tests, microbench A/B at 128 slots, default bench A/B
mkdir -p gpurun_out
python -m pytest tests/test_gpu_skinny_lds.py tests/test_gpu_llm.py -m gpu -x -q 2>&1 | grep -v amdgpu.ids | tail -15 > gpurun_out/fused_tests.log
python - >> gpurun_out/fused_tests.log 2>&1 <<'PY'
import torch
from aurora_amd import synthetic as S
from aurora_amd.engine import AuroraCapEngine, _rup
l = S.VICUNA_7B_16K
B, L0 = 128, 2142
eng = AuroraCapEngine({"vit": None, "llm": l}, {"llm": S.llm_weights(l)}, max_frames=1, max_batch=B, max_ctx=_rup(L0 + 256, 64), max_new_tokens=256)
eng.begin_batch(B, 256, None)
emb0 = (torch.randn(_rup(L0, 32), l["hidden_size"], device="cuda") * 0.02).half()
for b in range(B):
    eng.prefill(b, emb0.clone(), L0)
torch.cuda.synchronize()
for rep in range(2):
    for fused in (0, 1):
        eng.set_option("decode_fused_reduce", fused)
        print("fused", fused, {k: round(eng.microbench(k, 2000), 2) for k in ("dec_o", "dec_down")}, flush=True)
eng.set_option("decode_fused_reduce", 1)
import time
for fused in (0, 1, 0, 1):
    eng.set_option("decode_fused_reduce", fused)
    eng.decode(2)
    torch.cuda.synchronize()
    t = time.perf_counter()
    eng.decode(20)
    torch.cuda.synchronize()
    print("fused", fused, "decode step ms", round((time.perf_counter() - t) / 20 * 1e3, 3), flush=True)
    eng.begin_batch(B, 256, None)
    for b in range(B):
        eng.prefill(b, emb0.clone(), L0)
eng.close()
PY
for p in 1 0 1 0; do
  python bench.py --fused-reduce $p --no-cpu-baseline 2>/dev/null | tail -1 >> gpurun_out/fused_ab.jsonl
done
python - <<'PY' >> gpurun_out/fused_tests.log
import json
for l in open("gpurun_out/fused_ab.jsonl"):
    j = json.loads(l)
    print(j["value"], j["ms_per_step"], j.get("power", {}).get("sclk_mhz_p50"))
PY
cat gpurun_out/fused_tests.log
