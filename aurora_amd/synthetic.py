"""Synthetic AuroraCap-7B-shaped weights and inputs (no checkpoint / dataset is reachable offline).

Shapes follow the checkpoints named by the reference configs:
  visual encoder  apple/DFN5B-CLIP-ViT-H-14-378  (auroracap_7b_vision_stage.py:26-27)
  language model  lmsys/vicuna-7b-v1.5-16k        (auroracap_7b_language_stage.py:37; RoPE linear x4)
Recipe: Linear / conv / embedding weights N(0, 0.02^2), LayerNorm / RMSNorm weight 1, bias 0,
fp16, seeded.  Frames: uint8 U[0,255], CLIP mean/std normalisation (inference.py:58-63 processor).
`norm_std` / `bias_std` > 0 (parity tests only; the benchmark keeps the recipe above) draw the norm weights as
1 + N(0, norm_std^2) and every bias as N(0, bias_std^2) from a SEPARATE generator, so the matrices stay those of the
recipe: a real checkpoint's norm weights are far from 1, and the decode path folds them into the projections.
"""
from __future__ import annotations

import numpy as np
import torch

VIT_H_378 = dict(hidden_size=1280, num_attention_heads=16, num_hidden_layers=32, intermediate_size=5120, patch_size=14,
                 image_size=378, num_channels=3, hidden_act="quick_gelu", layer_norm_eps=1e-5)
VICUNA_7B_16K = dict(hidden_size=4096, num_attention_heads=32, num_hidden_layers=32, intermediate_size=11008,
                     vocab_size=32000, rms_norm_eps=1e-5, rope_theta=10000.0, rope_factor=4.0)
AURORACAP_7B = {"vit": VIT_H_378, "llm": VICUNA_7B_16K}

CLIP_MEAN = (0.48145466, 0.4578275, 0.40821073)
CLIP_STD = (0.26862954, 0.26130258, 0.27577711)
IMAGE_TOKEN_INDEX = -200


def _rn(gen, *shape, std=0.02, device="cuda"):
    return (torch.randn(*shape, generator=gen, device=device, dtype=torch.float32) * std).to(torch.float16)


def _norm_like(gen, n, mean, std, device):
    if std <= 0:
        return torch.full((n,), mean, dtype=torch.float16, device=device)
    return (mean + torch.randn(n, generator=gen, device=device, dtype=torch.float32) * std).to(torch.float16)


def vit_weights(cfg, seed=1234, device="cuda", num_layers=None, norm_std=0.0, bias_std=0.0):
    g = torch.Generator(device=device).manual_seed(seed)
    g2 = torch.Generator(device=device).manual_seed(seed + 7777)
    D, mlp, P, C = cfg["hidden_size"], cfg["intermediate_size"], cfg["patch_size"], cfg.get("num_channels", 3)
    t0 = (cfg["image_size"] // P) ** 2 + 1
    one = lambda n: _norm_like(g2, n, 1.0, norm_std, device)
    zero = lambda n: _norm_like(g2, n, 0.0, bias_std, device)
    w = {"patch_embedding.weight": _rn(g, D, C, P, P, device=device), "class_embedding": _rn(g, D, device=device),
         "position_embedding.weight": _rn(g, t0, D, device=device), "pre_layrnorm.weight": one(D), "pre_layrnorm.bias": zero(D),
         "layers": []}
    for _ in range(cfg["num_hidden_layers"] if num_layers is None else num_layers):
        lw = {}
        for n in ("q_proj", "k_proj", "v_proj", "out_proj"):
            lw[n + ".weight"], lw[n + ".bias"] = _rn(g, D, D, device=device), zero(D)
        lw["fc1.weight"], lw["fc1.bias"] = _rn(g, mlp, D, device=device), zero(mlp)
        lw["fc2.weight"], lw["fc2.bias"] = _rn(g, D, mlp, device=device), zero(D)
        for n in ("layer_norm1", "layer_norm2"):
            lw[n + ".weight"], lw[n + ".bias"] = one(D), zero(D)
        w["layers"].append(lw)
    return w


def projector_weights(dv, d, seed=1235, device="cuda"):
    g = torch.Generator(device=device).manual_seed(seed)
    z = lambda n: torch.zeros(n, dtype=torch.float16, device=device)
    return {"model.0.weight": _rn(g, d, dv, device=device), "model.0.bias": z(d),
            "model.2.weight": _rn(g, d, d, device=device), "model.2.bias": z(d)}


def llm_weights(cfg, seed=1236, device="cuda", num_layers=None, norm_std=0.0):
    g = torch.Generator(device=device).manual_seed(seed)
    g2 = torch.Generator(device=device).manual_seed(seed + 7777)
    d, mlp, V = cfg["hidden_size"], cfg["intermediate_size"], cfg["vocab_size"]
    one = lambda n: _norm_like(g2, n, 1.0, norm_std, device)
    w = {"embed_tokens.weight": _rn(g, V, d, device=device), "norm.weight": one(d), "lm_head.weight": _rn(g, V, d, device=device),
         "layers": []}
    for _ in range(cfg["num_hidden_layers"] if num_layers is None else num_layers):
        lw = {n + ".weight": _rn(g, d, d, device=device) for n in ("q_proj", "k_proj", "v_proj", "o_proj")}
        lw["gate_proj.weight"], lw["up_proj.weight"] = _rn(g, mlp, d, device=device), _rn(g, mlp, d, device=device)
        lw["down_proj.weight"] = _rn(g, d, mlp, device=device)
        lw["input_layernorm.weight"], lw["post_attention_layernorm.weight"] = one(d), one(d)
        w["layers"].append(lw)
    return w


def frames(num_frm, clip_id, image=378, device="cuda"):
    """uint8 U[0,255] frames [f, H, W, 3] seeded 1000 + clip_id -> normalised fp16 [f, 3, H, W]."""
    rng = np.random.default_rng(1000 + clip_id)
    u8 = rng.integers(0, 256, size=(num_frm, image, image, 3), dtype=np.uint8)
    x = torch.from_numpy(u8).to(device).float() / 255.0
    mean = torch.tensor(CLIP_MEAN, device=device)
    std = torch.tensor(CLIP_STD, device=device)
    return ((x - mean) / std).permute(0, 3, 1, 2).contiguous().to(torch.float16)


def prompt_ids(num_frm, clip_id, n_text=30, vocab=32000):
    """BOS + random text ids with num_frm IMAGE_TOKEN_INDEX markers placed as inference.py:76-86 does
    ("USER: <image> <image> ... \\n{prompt} ASSISTANT:"): a short prefix, marker / separator pairs, the rest."""
    rng = np.random.default_rng(2000 + clip_id)
    text = rng.integers(3, vocab, size=n_text - 1).tolist()
    ids = [1] + text[:3]
    rest = text[3:]
    for i in range(num_frm):
        ids.append(IMAGE_TOKEN_INDEX)
        if i != num_frm - 1:
            ids.append(rest.pop(0))
    ids.extend(rest)
    return ids
