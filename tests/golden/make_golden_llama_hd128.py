#!/usr/bin/env python3
"""G9b: the decode model of the path - HF `LlamaForCausalLM` as loaded at /root/reference/inference.py:47-51
(`AutoModelForCausalLM.from_pretrained`, lmsys/vicuna-7b-v1.5-16k per auroracap_7b_language_stage.py:26: RoPE "linear" scaling x4) -
at the REAL head_dim (128) and at the positions the BASELINE configs reach (cfg2 prefill ends at 2142, cfg5's context at 6.9 k).

G9 (make_golden.py) pins the restatement on hidden 64 / head_dim 16 / 48 positions; this fixture pins `oracle.aurora_oracle.llama_forward`
- inverse frequencies theta^(-2i/128), position / 4, cos / sin in fp32, half-split rotation, causal mask, RMSNorm, SwiGLU - where the
7B-width GPU tests use it: one sequence of 6874 token ids through a 2-layer, 2-head x 128 model, logits kept at rows 0..7, 2100..2107
and 6850..6873 (so the fixture stays small: the input is ids, the weights are fp16-representable and stored as fp16).

Third-party arithmetic: transformers (container: 5.15.0; the reference pins <= 4.42.4 - drift recorded in `versions`).
Re-run:  python tests/golden/make_golden_llama_hd128.py      (CPU, ~10 s; needs transformers, not /root/reference)
"""
import json
import os

import numpy as np
import torch

HERE = os.path.dirname(os.path.abspath(__file__))
ROWS = list(range(0, 8)) + list(range(2100, 2108)) + list(range(6850, 6874))
T = 6874


def main():
    import transformers
    from transformers import LlamaConfig, LlamaForCausalLM
    torch.manual_seed(1234)
    lcfg = dict(vocab_size=128, hidden_size=256, intermediate_size=512, num_hidden_layers=2, num_attention_heads=2,
                num_key_value_heads=2, max_position_embeddings=16384, rms_norm_eps=1e-5, rope_theta=1e4, bos_token_id=1,
                eos_token_id=2, pad_token_id=0, tie_word_embeddings=False)
    try:
        cfg = LlamaConfig(**lcfg, rope_scaling={"type": "linear", "factor": 4.0})
    except Exception:
        cfg = LlamaConfig(**lcfg, rope_scaling={"rope_type": "linear", "factor": 4.0})
    model = LlamaForCausalLM(cfg).eval()
    with torch.no_grad():
        for n, p in model.named_parameters():
            if "norm" in n:
                p.copy_(1.0 + 0.1 * torch.randn_like(p))
            elif "embed_tokens" in n:
                p.copy_(torch.randn_like(p))                      # unit-scale residual stream, like a trained model's
            else:
                p.copy_(torch.randn_like(p) * (0.6 / p.shape[1] ** 0.5))     # attention logits of order 1: RoPE phases matter
            p.copy_(p.half().float())                             # fp16-representable: the fixture stores fp16, nothing is lost
    assert model.config.hidden_size // model.config.num_attention_heads == 128
    ids = torch.randint(3, 128, (1, T))
    with torch.no_grad():
        out = model(input_ids=ids)                                # position_ids default to 0 .. T-1, as generate() passes them (utils.py:280-295 passes none)
        logits = out.logits[0, ROWS].float()
        # the same rows through HF's own KV-cache path: prefill 6850 positions, then one position per call (what generate does)
        pre = model(input_ids=ids[:, :6850], use_cache=True)
        past = pre.past_key_values
        step_logits = []
        for t in range(6850, T):
            o = model(input_ids=ids[:, t:t + 1], past_key_values=past, use_cache=True)
            past = o.past_key_values
            step_logits.append(o.logits[0, 0].float())
        step_logits = torch.stack(step_logits)
    cache_vs_full = (step_logits - logits[16:]).abs().max().item()
    assert cache_vs_full < 1e-3, cache_vs_full                     # HF's two paths agree; the full-sequence logits are the fixture
    fx = {f"w.{k}": v.detach().half().numpy() for k, v in model.state_dict().items() if "rotary" not in k and "inv_freq" not in k}
    fx.update(ids=ids[0].numpy().astype(np.int16), rows=np.asarray(ROWS, np.int32), logits=logits.numpy(),
              hf_cache_vs_full_max_abs=np.float32(cache_vs_full),
              cfg=np.array(json.dumps(dict(lcfg, rope_factor=4.0))),
              versions=np.array(json.dumps({"torch": torch.__version__, "transformers": transformers.__version__,
                                            "reference_pins": "transformers>=4.36.0,!=4.38.0-2,<=4.42.4 (src/xtuner/requirements/runtime.txt:26)"})))
    path = os.path.join(HERE, "g9b_llama_hd128_long.npz")
    np.savez(path, **fx)
    print(path, os.path.getsize(path) / 1e6, "MB; logit scale", logits.abs().max().item(), "cache vs full", cache_vs_full)


if __name__ == "__main__":
    main()
