"""A/B of single kernels between two builds of the library on one box:  AURORA_HIP_SO=<path> python tools/gpu/ab_microbench.py
Prints microbench times (us) of the prefill / decode kernels at the default bench's shapes (128 slots, context 2142, prefill groups of 4),
on all CUs and on the serving schedule's CU masks."""
import os
import sys

import torch

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
from aurora_amd import synthetic as S                     # noqa: E402
from aurora_amd.engine import AuroraCapEngine, _rup       # noqa: E402
from aurora_amd.streams import shared_cu_masked_stream    # noqa: E402

l = S.VICUNA_7B_16K
B, L0 = 128, 2142
eng = AuroraCapEngine({"vit": None, "llm": l}, {"llm": S.llm_weights(l)}, max_frames=1, max_batch=B, max_ctx=_rup(L0 + 256, 64), max_new_tokens=256)
eng.begin_batch(B, 256, None)
emb0 = (torch.randn(_rup(L0, 32), l["hidden_size"], device="cuda") * 0.02).half()
for b in range(B):
    eng.prefill(b, emb0.clone(), L0)
torch.cuda.synchronize()
eng.set_option("microbench_prefill_nseq", 4)
sf, sd = shared_cu_masked_stream(16), shared_cu_masked_stream(16, from_top=True)
names = sys.argv[1:] or ["pre_attn", "dec_attn"]
for rep in range(2):
    out = {}
    for k in names:
        it = 300 if k.startswith("pre") else 600
        out[k] = round(eng.microbench(k, it), 1)
        with torch.cuda.stream(sf if k.startswith("pre") else sd):
            out[k + "@16cu"] = round(eng.microbench(k, it), 1)
    print(os.environ.get("AURORA_HIP_SO", "default"), out, flush=True)
eng.close()
