"""Checkpoint loading for the xtuner-format AuroraCap directory (host-side housekeeping).

Layout written by the reference (aurora.py:312-362 `to_xtuner_llava`, consumed at inference.py:42-57):
    <root>/                 HF Llama (config.json + *.safetensors | pytorch_model*.bin) + tokenizer files
    <root>/visual_encoder/  HF CLIP vision tower  (keys vision_model.*; `visual_encoder.pos_emb` aliases
                            position_embedding.weight, aurora.py:878 / pth_to_hf.py:119-124)
    <root>/projector/       xtuner ProjectorModel (keys model.0.*, model.2.*)
Everything architectural is read from the config.json files - never hard-coded.
Returns (cfg, weights) in the naming of aurora_amd.engine / oracle.aurora_oracle.
"""
from __future__ import annotations

import glob
import json
import os
from typing import Dict, Tuple

import torch


def _load_state(d: str) -> Dict[str, torch.Tensor]:
    """All tensors of one `save_pretrained` directory: safetensors or torch-pickle files, single or sharded.  A shard index
    (`*.index.json`, written next to sharded weights) is authoritative: exactly the files its weight_map names are read and
    every key it lists must turn up - a directory holding stale extra weight files is then still read correctly."""
    sd: Dict[str, torch.Tensor] = {}
    for pattern, idx_name in (("*.safetensors", "model.safetensors.index.json"), ("pytorch_model*.bin", "pytorch_model.bin.index.json")):
        files = sorted(glob.glob(os.path.join(d, pattern)))
        idx = os.path.join(d, idx_name)
        wmap = _json(idx)["weight_map"] if os.path.exists(idx) else None
        if wmap is not None:
            files = [os.path.join(d, f) for f in sorted(set(wmap.values()))]
        if not files:
            continue
        for f in files:
            if f.endswith(".safetensors"):
                from safetensors.torch import load_file
                sd.update(load_file(f))
            else:
                sd.update(torch.load(f, map_location="cpu", weights_only=True))
        if wmap is not None:
            missing = [k for k in wmap if k not in sd]
            if missing:
                raise KeyError(f"{idx_name} lists tensors its shards do not hold: {missing[:4]}")
        return sd
    raise FileNotFoundError(f"no *.safetensors or pytorch_model*.bin under {d}")


def _json(p):
    with open(p) as f:
        return json.load(f)


def vit_config(d: str) -> dict:
    c = _json(os.path.join(d, "config.json"))
    c = c.get("vision_config", c)
    return dict(hidden_size=c["hidden_size"], num_attention_heads=c["num_attention_heads"],
                num_hidden_layers=c["num_hidden_layers"], intermediate_size=c["intermediate_size"],
                patch_size=c["patch_size"], image_size=c["image_size"], num_channels=c.get("num_channels", 3),
                hidden_act=c.get("hidden_act", "quick_gelu"), layer_norm_eps=c.get("layer_norm_eps", 1e-5))


def llm_config(d: str) -> dict:
    c = _json(os.path.join(d, "config.json"))
    if c.get("num_key_value_heads", c["num_attention_heads"]) != c["num_attention_heads"]:
        raise NotImplementedError("grouped-query attention is not on the AuroraCap-7B path (vicuna-7b is MHA)")
    # transformers <= 4.x (the reference's pin; vicuna-7b-v1.5-16k): "rope_scaling": {"type": "linear", "factor": 4.0} + "rope_theta";
    # transformers 5.x writes ONE dict instead: "rope_parameters": {"rope_type": "linear", "factor": 4.0, "rope_theta": ...}
    rp = c.get("rope_parameters") or {}
    rs = c.get("rope_scaling") or {k: v for k, v in rp.items() if k != "rope_theta"}
    if set(rs) <= {"rope_type", "type"} and rs.get("rope_type", rs.get("type")) in (None, "default"):
        rs = {}
    kind = rs.get("type", rs.get("rope_type", "linear" if rs else None))
    if rs and kind not in ("linear", "default", None):
        raise NotImplementedError(f"rope_scaling type {kind!r} (vicuna-7b-v1.5-16k uses linear)")
    theta = c.get("rope_theta", rp.get("rope_theta", 10000.0))
    return dict(hidden_size=c["hidden_size"], num_attention_heads=c["num_attention_heads"],
                num_hidden_layers=c["num_hidden_layers"], intermediate_size=c["intermediate_size"],
                vocab_size=c["vocab_size"], rms_norm_eps=c.get("rms_norm_eps", 1e-5), rope_theta=theta,
                rope_factor=float(rs.get("factor", 1.0)) if kind == "linear" else 1.0,
                eos_token_id=c.get("eos_token_id", 2), bos_token_id=c.get("bos_token_id", 1))


def projector_config(d: str) -> dict:
    """projector/config.json as the reference's `ProjectorConfig` reads it (configuration_projector.py:9-22): absent keys take the
    class defaults.  `ProjectorModel.__init__` (modeling_projector.py:20-33) builds `depth` Linear layers with `ACT2FN[hidden_act]`
    between them and an optional bias - all three are honoured by the engine or rejected loudly (aurora_amd.engine.projector_settings)."""
    p = os.path.join(d, "config.json")
    if not os.path.exists(p):      # the reference's ProjectorModel.from_pretrained needs it too; the class defaults (4096 -> 4096) fit no AuroraCap
        raise FileNotFoundError(f"{p} not found: the projector sub-directory of an xtuner-format checkpoint holds config.json beside its weights")
    c = _json(p)
    return dict(visual_hidden_size=c.get("visual_hidden_size", 4096), llm_hidden_size=c.get("llm_hidden_size", 4096),
                depth=c.get("depth", 2), hidden_act=c.get("hidden_act", "gelu"), bias=c.get("bias", True))


def projector_weights(sd: Dict[str, torch.Tensor], pcfg: dict) -> dict:
    """`model.0`, `model.2`, .. `model.{2 (depth - 1)}` (the odd positions of the nn.Sequential are the activations); every key the
    config implies must be there and nothing else"""
    want = [f"model.{2 * i}.{n}" for i in range(int(pcfg["depth"])) for n in (("weight", "bias") if pcfg["bias"] else ("weight",))]
    missing = [k for k in want if k not in sd]
    extra = [k for k in sd if k.startswith("model.") and k not in want]
    if missing or extra:
        raise KeyError(f"projector checkpoint does not match its config (depth {pcfg['depth']}, bias {pcfg['bias']}): missing {missing}, unexpected {extra}")
    return {k: sd[k] for k in want}


def vit_weights(sd: Dict[str, torch.Tensor], cfg: dict) -> dict:
    p = "vision_model."
    if not any(k.startswith(p) for k in sd):
        p = ""
    g = lambda k: sd[p + k]
    w = {"patch_embedding.weight": g("embeddings.patch_embedding.weight"), "class_embedding": g("embeddings.class_embedding"),
         # the table the reference's forward really uses is its `pos_emb` alias (aurora.py:878, 899); a checkpoint saved through
         # safetensors cannot hold both names of one tensor and keeps either - take whichever is there
         "position_embedding.weight": sd["pos_emb"] if "pos_emb" in sd else g("embeddings.position_embedding.weight"),
         "pre_layrnorm.weight": g("pre_layrnorm.weight"), "pre_layrnorm.bias": g("pre_layrnorm.bias"), "layers": []}
    for i in range(cfg["num_hidden_layers"]):
        q = f"encoder.layers.{i}."
        lw = {}
        for n in ("q_proj", "k_proj", "v_proj", "out_proj"):
            lw[n + ".weight"], lw[n + ".bias"] = g(q + f"self_attn.{n}.weight"), g(q + f"self_attn.{n}.bias")
        for n in ("fc1", "fc2"):
            lw[n + ".weight"], lw[n + ".bias"] = g(q + f"mlp.{n}.weight"), g(q + f"mlp.{n}.bias")
        for n in ("layer_norm1", "layer_norm2"):
            lw[n + ".weight"], lw[n + ".bias"] = g(q + n + ".weight"), g(q + n + ".bias")
        w["layers"].append(lw)
    return w


def llm_weights(sd: Dict[str, torch.Tensor], cfg: dict) -> dict:
    w = {"embed_tokens.weight": sd["model.embed_tokens.weight"], "norm.weight": sd["model.norm.weight"],
         "lm_head.weight": sd.get("lm_head.weight", sd["model.embed_tokens.weight"]), "layers": []}
    for i in range(cfg["num_hidden_layers"]):
        q = f"model.layers.{i}."
        lw = {n + ".weight": sd[q + f"self_attn.{n}.weight"] for n in ("q_proj", "k_proj", "v_proj", "o_proj")}
        for n in ("gate_proj", "up_proj", "down_proj"):
            lw[n + ".weight"] = sd[q + f"mlp.{n}.weight"]
        lw["input_layernorm.weight"] = sd[q + "input_layernorm.weight"]
        lw["post_attention_layernorm.weight"] = sd[q + "post_attention_layernorm.weight"]
        w["layers"].append(lw)
    return w


def load_auroracap(root: str, visual_encoder: str = "visual_encoder", projector: str = "projector") -> Tuple[dict, dict]:
    """(cfg, weights) from an xtuner-format AuroraCap directory (inference.py:42-45: <root>, <root>/visual_encoder,
    <root>/projector; the sub-directory names can be overridden)."""
    vdir, pdir = os.path.join(root, visual_encoder), os.path.join(root, projector)
    for d in (root, vdir, pdir):
        if not os.path.isdir(d):
            raise FileNotFoundError(f"{d} is not a directory (expected <root>/, <root>/visual_encoder, <root>/projector)")
    cfg = {"vit": vit_config(vdir), "llm": llm_config(root), "projector": projector_config(pdir)}
    for key, have in (("visual_hidden_size", cfg["vit"]["hidden_size"]), ("llm_hidden_size", cfg["llm"]["hidden_size"])):
        if cfg["projector"][key] != have:
            raise ValueError(f"projector/config.json {key} = {cfg['projector'][key]} but the loaded model has {have}")
    weights = {"vit": vit_weights(_load_state(vdir), cfg["vit"]), "llm": llm_weights(_load_state(root), cfg["llm"]),
               "projector": projector_weights(_load_state(pdir), cfg["projector"])}
    return cfg, weights
