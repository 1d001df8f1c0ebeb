#!/usr/bin/env python3
"""The decode attention kernel at the metric's shape (round 6 used it to A/B two page pipelines - profiles/r06_attn_pipeline_ab.log; the losing one is gone): 128 slots x 32 heads, context 2142,
on the whole chip and on a stream that owns 16 CUs of every XCD (the masked decode steps of the serving loop).

    python tools/gpu/attn_ab.py [--batch 128] [--ctx 2142]
"""
import argparse
import json
import os
import sys

import torch

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--batch", type=int, default=128)
    ap.add_argument("--ctx", type=int, default=2142)
    ap.add_argument("--iters", type=int, default=320)
    args = ap.parse_args()
    from aurora_amd import synthetic as S
    from aurora_amd.engine import AuroraCapEngine, _rup
    from aurora_amd.streams import cu_masked_stream
    l = S.VICUNA_7B_16K
    B, L0 = args.batch, args.ctx
    eng = AuroraCapEngine({"vit": None, "llm": l}, {"llm": S.llm_weights(l)}, max_frames=1, max_batch=B, max_ctx=_rup(L0 + 256, 64), max_new_tokens=256)
    torch.cuda.empty_cache()
    d = l["hidden_size"]
    eng.begin_batch(B, 256, None)
    g = torch.Generator(device="cuda").manual_seed(0)
    emb0 = (torch.randn(_rup(L0, 32), d, generator=g, device="cuda") * 0.02).half()
    for b in range(B):
        eng.prefill(b, emb0.clone(), L0)
    torch.cuda.synchronize()
    kv_bytes = B * (L0 + 1) * 2 * d * 2
    res = {}
    base = torch.cuda.current_stream()
    for cus in (0, 16):
        if cus:
            torch.cuda.synchronize()
            torch.cuda.set_stream(cu_masked_stream(cus, from_top=True))
        for rep in range(2):
            for v in (1,):
                us = eng.microbench("dec_attn", args.iters)
                key = f"cus={cus or 32} rep={rep}"
                res[key] = (round(us, 1), round(kv_bytes / us / 1e6, 3))
                print(f"{key:32s} {us:8.1f} us  {kv_bytes / us / 1e6:6.3f} TB/s", flush=True)
    torch.cuda.synchronize()
    torch.cuda.set_stream(base)
    print(json.dumps(res))
    eng.close()


if __name__ == "__main__":
    main()
