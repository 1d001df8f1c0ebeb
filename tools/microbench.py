#!/usr/bin/env python3
"""Per-kernel micro-benchmarks of the LLM half of the path at AuroraCap-7B dims (aur_microbench).

    python tools/microbench.py --batch 8 [--ctx 2142]

Prints mean microseconds per launch and the achieved algorithmic GB/s (HBM-bound kernels) or TFLOP/s
(MFMA-bound kernels) for a few tuning-knob settings.  Launches cycle through all 32 layers, so weights
stream from HBM exactly as in a real decode step.
"""
import argparse
import json
import os
import sys

import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--batch", type=int, default=8)
    ap.add_argument("--ctx", type=int, default=2142)
    ap.add_argument("--iters", type=int, default=320)
    ap.add_argument("--only", type=str, default="")
    ap.add_argument("--quick", action="store_true")
    ap.add_argument("--nseq", type=int, default=1, help="sequences per prefill pass for the pre_* kernels (<= batch)")
    ap.add_argument("--ab-option", type=str, default="", help="A/B an engine option (0 / 1 / 0 / 1) on the prefill projections, e.g. gemm_nt_out")
    ap.add_argument("--half-grid", action="store_true", help="decode projections with half as many workgroups (decode_half_grid)")
    ap.add_argument("--mask-cus", type=int, default=0, help="time the kernels on a stream restricted to this many CUs of every XCD")
    args = ap.parse_args()
    from aurora_amd import synthetic as S
    from aurora_amd.engine import AuroraCapEngine, _rup
    l = S.VICUNA_7B_16K
    B, L0 = args.batch, args.ctx
    eng = AuroraCapEngine({"vit": None, "llm": l}, {"llm": S.llm_weights(l)}, max_frames=1, max_batch=B,
                          max_ctx=_rup(L0 + 256, 64), max_new_tokens=256)
    torch.cuda.empty_cache()
    d, mlp, V, H = l["hidden_size"], l["intermediate_size"], l["vocab_size"], l["num_attention_heads"]
    eng.begin_batch(B, 256, None)
    g = torch.Generator(device="cuda").manual_seed(0)
    for b in range(B):
        emb = (torch.randn(_rup(L0, 32), d, generator=g, device="cuda") * 0.02).half()
        eng.prefill(b, emb, L0)
    torch.cuda.synchronize()
    if args.mask_cus > 0:
        from aurora_amd.streams import cu_masked_stream
        torch.cuda.set_stream(cu_masked_stream(args.mask_cus, from_top=True))
        eng.set_option("gemm_max_wgs", 8 * args.mask_cus)
    if args.half_grid:
        eng.set_option("decode_half_grid", 1)
    eng.set_option("microbench_prefill_nseq", min(args.nseq, B))
    M = _rup(L0, 32) * min(args.nseq, B)
    kv_bytes = B * (L0 + 1) * 2 * d * 2
    work = {   # kernel -> (unit, amount per launch)
        "dec_qkv": ("GB/s", 3 * d * d * 2), "dec_o": ("GB/s", d * d * 2), "dec_gateup": ("GB/s", 2 * mlp * d * 2),
        "dec_down": ("GB/s", mlp * d * 2), "dec_lm_head": ("GB/s", V * d * 2), "dec_attn": ("GB/s", kv_bytes),
        "pre_qkv": ("TF/s", 2 * M * 3 * d * d), "pre_o": ("TF/s", 2 * M * d * d), "pre_gateup": ("TF/s", 2 * M * 2 * mlp * d),
        "pre_down": ("TF/s", 2 * M * mlp * d), "pre_attn": ("TF/s", 2 * L0 * L0 * d * min(args.nseq, B)), "pre_norm": ("GB/s", 2 * M * d * 2),
    }
    res = {}

    def run(tag, kernels):
        for k in kernels:
            if args.only and args.only not in k:
                continue
            us = eng.microbench(k, args.iters if k.startswith("dec") else 64)
            unit, amt = work[k]
            rate = amt / us / (1e3 if unit == "GB/s" else 1e6)
            res[f"{tag}:{k}"] = (round(us, 2), round(rate, 1), unit)
            print(f"{tag:28s} {k:12s} {us:9.2f} us  {rate:9.1f} {unit}", flush=True)

    dec = ["dec_qkv", "dec_o", "dec_gateup", "dec_down", "dec_lm_head"]
    run("decode projections (the engine's structure for %d slots)" % B, dec)
    if not args.quick:
        run("attn (engine's splits)", ["dec_attn"])
        for pps in (1, 2, 4, 8, 19, 38):
            eng.set_option("dec_attn_pps", pps)
            run(f"attn pps{pps}", ["dec_attn"])
    else:
        run("attn (engine's splits)", ["dec_attn"])
    for mode, tag in ((0, "prefill gemm128"), (2, "prefill gemm256")):
        eng.set_option("gemm_mode", mode)
        run(tag, ["pre_qkv", "pre_o", "pre_gateup", "pre_down"])
    eng.set_option("gemm_mode", 1)
    if args.ab_option:
        for v in (0, 1, 0, 1):
            eng.set_option(args.ab_option, v)
            run(f"{args.ab_option} = {v}", ["pre_qkv", "pre_o", "pre_gateup", "pre_down"])
    run("prefill", ["pre_norm", "pre_attn"])
    print(json.dumps(res))
    eng.close()


if __name__ == "__main__":
    main()
