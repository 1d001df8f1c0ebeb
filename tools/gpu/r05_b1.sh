#!/bin/bash
mkdir -p gpurun_out/r05
run() { python bench.py --batch ${B:-1} --steps 2 --warmup 1 --no-cpu-baseline --no-instrument "$@" 2>/dev/null | python -c "
import json,sys
d=json.loads(sys.stdin.read().strip().splitlines()[-1]); ds=d['decode_step']
print('$*', ':', round(d['value'],4), 'captions/s, decode', round(ds['ms_per_step'],4), 'ms/step, frac', round(ds['frac'],4))"; }
run
run --dec-attn-pps 2
run --dec-attn-pps 1
run --dec-attn-pps 4
B=8 run
B=8 run --dec-attn-pps 10
python - <<'PY'
import sys, torch
sys.path.insert(0, '.')
from aurora_amd import synthetic as S
from aurora_amd.engine import AuroraCapEngine, _rup
l = S.VICUNA_7B_16K
for B in (1, 8):
    eng = AuroraCapEngine({"vit": None, "llm": l}, {"llm": S.llm_weights(l)}, max_frames=1, max_batch=B, max_ctx=_rup(2142 + 256, 64), max_new_tokens=256)
    eng.begin_batch(B, 256, None)
    g = torch.Generator(device="cuda").manual_seed(0)
    for b in range(B):
        eng.prefill(b, (torch.randn(_rup(2142, 32), 4096, generator=g, device="cuda") * 0.02).half(), 2142)
    torch.cuda.synchronize()
    for nw in (8, 16):
        eng.set_option("dec_row_waves", nw)
        print(B, "slots, row waves", nw, {k: round(eng.microbench(k, 320), 2) for k in ("dec_qkv", "dec_o", "dec_gateup", "dec_down", "dec_attn")})
    eng.close()
PY
