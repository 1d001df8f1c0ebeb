"""CPU oracle for the AuroraCap inference hot path.  TEST INFRASTRUCTURE ONLY.

This file is a plain PyTorch-CPU fp32 restatement of the reference algorithm
(rese1f/aurora @ 2025-06-14).  Only ``tests/``, ``__graft_entry__.smoke()`` and
``bench.py``'s ``cpu_baseline`` leg may import it; the product path
(``aurora_amd``) never does and fails loudly when its HIP library is missing.

Parity pin: every function below is checked in ``tests/test_oracle_golden.py``
against golden vectors produced by importing the reference's own modules in the
build container (``tests/golden/make_golden.py``), because the reference ships
no tests of its own for this path.  The Llama decode
arithmetic lives in third-party ``transformers`` (reference pin
``>=4.36,<=4.42.4``, src/xtuner/requirements/runtime.txt:26; fixtures were made
with transformers 5.15.0, the only version in the container - drift noted).

Each function cites the reference file:line it follows (paths relative to the
reference root).
"""
from __future__ import annotations

import math
from typing import Callable, Dict, List, Optional, Sequence, Tuple

import torch
import torch.nn.functional as F

IMAGE_TOKEN_INDEX = -200          # src/xtuner/xtuner/utils/constants.py:4
DEFAULT_IMAGE_TOKEN = "<image>"   # src/xtuner/xtuner/utils/constants.py:5
VICUNA_INSTRUCTION = "USER: {input} ASSISTANT:"  # src/xtuner/xtuner/utils/templates.py:92

Q = Optional[Callable[[torch.Tensor], torch.Tensor]]


def _id(x):
    return x


def fp16_storage(x: torch.Tensor) -> torch.Tensor:
    """Rounding hook emulating the kernels' fp16 storage points (fp32 math)."""
    return x.to(torch.float16).to(torch.float32)


# --------------------------------------------------------------------------
# token schedule  (aurora.py:895, tome.py:45, aurora.py:253)
# --------------------------------------------------------------------------
def tome_r(height: int, width: int, patch: int, token_kept_ratio: float, num_layers: int) -> int:
    """aurora.py:895 - evaluated in Python doubles in exactly this order."""
    return int(width * height / (patch ** 2) * (1 - token_kept_ratio) / num_layers)


def token_schedule(t0: int, r: int, num_layers: int) -> List[int]:
    """Token count entering each layer 0..L (index L = after the last layer).

    tome.py:45 clamps r to (t - 1) // 2 (class token protected); r <= 0 is identity.
    """
    ts = [t0]
    t = t0
    for _ in range(num_layers):
        rl = min(r, (t - 1) // 2)
        if rl > 0:
            t -= rl
        ts.append(t)
    return ts


def kept_tokens(height: int, width: int, patch: int, ratio: float, num_layers: int) -> int:
    """Visual tokens per frame consumed by the LLM: hidden_states[-2][:, 1:] (aurora.py:253)."""
    t0 = (height // patch) * (width // patch) + 1
    r = tome_r(height, width, patch, ratio, num_layers)
    return token_schedule(t0, r, num_layers)[num_layers - 1] - 1


# --------------------------------------------------------------------------
# ToMe  (tome.py:18-98, 207-219)
# --------------------------------------------------------------------------
def bipartite_match(metric: torch.Tensor, r: int):
    """tome.py:36-69 with class_token=True.  metric [n, t, c] -> index tensors.

    Returns None when the step is the identity (r <= 0 after clamping), else a
    dict with node_max [n, ta], node_idx [n, ta], unm_idx [n, ta-r] (ascending),
    src_idx [n, r] (by descending node_max, stable), dst_idx [n, r].
    """
    t = metric.shape[1]
    r = min(r, (t - 1) // 2)
    if r <= 0:
        return None
    m = metric / metric.norm(dim=-1, keepdim=True)
    a, b = m[..., ::2, :], m[..., 1::2, :]
    scores = a @ b.transpose(-1, -2)
    scores[..., 0, :] = -math.inf
    node_max, node_idx = scores.max(dim=-1)
    edge_idx = node_max.argsort(dim=-1, descending=True, stable=True)
    unm_idx = edge_idx[..., r:].sort(dim=-1)[0]
    src_idx = edge_idx[..., :r]
    dst_idx = node_idx.gather(dim=-1, index=src_idx)
    return dict(r=r, node_max=node_max, node_idx=node_idx, unm_idx=unm_idx,
                src_idx=src_idx, dst_idx=dst_idx)


def _merge_sum(x: torch.Tensor, match) -> torch.Tensor:
    """tome.py:71-81 merge(mode='sum'): cat([A[unm], B.scatter_add(dst, A[src])])."""
    src, dst = x[..., ::2, :], x[..., 1::2, :]
    n, _, c = src.shape
    r = match["r"]
    unm_idx, src_idx, dst_idx = match["unm_idx"], match["src_idx"], match["dst_idx"]
    unm = src.gather(dim=-2, index=unm_idx[..., None].expand(n, unm_idx.shape[1], c))
    srcv = src.gather(dim=-2, index=src_idx[..., None].expand(n, r, c))
    dst = dst.scatter_reduce(-2, dst_idx[..., None].expand(n, r, c), srcv, reduce="sum")
    return torch.cat([unm, dst], dim=1)


def merge_wavg(x: torch.Tensor, size: Optional[torch.Tensor], match):
    """tome.py:207-219.  x [n,t,c], size [n,t,1] or None -> (x', size')."""
    if size is None:
        size = torch.ones_like(x[..., 0, None])
    if match is None:
        return x, size
    xs = _merge_sum(x * size, match)
    size = _merge_sum(size, match)
    return xs / size, size


def tome_step(metric, x, size, r):
    """aurora.py:746-747: one ToMe step for all frames at one layer."""
    match = bipartite_match(metric, r)
    x2, size2 = merge_wavg(x, size, match)
    return x2, size2, match


# --------------------------------------------------------------------------
# ViT  (HF CLIP <=4.42 wrapper semantics called at aurora.py:899; aurora.py:600-860)
# --------------------------------------------------------------------------
def quick_gelu(x):
    return x * torch.sigmoid(1.702 * x)


ACT = {"quick_gelu": quick_gelu, "gelu": lambda x: F.gelu(x)}


def interpolate_pos_encoding(pos: torch.Tensor, height: int, width: int, patch: int) -> torch.Tensor:
    """aurora.py:909-951: position embedding [1 + N, D] for an input of height x width pixels.  Unchanged when the
    patch grid is the native square one (:919-924); otherwise the N = n x n patch rows are resampled bicubically to
    (height // patch, width // patch) with scale factors (g + 0.1) / n (:933-940), the class row is kept."""
    w0, h0 = height // patch, width // patch                              # the reference's names (:915-916)
    n_tok = pos.shape[0] - 1
    if w0 * h0 == n_tok and w0 == h0:
        return pos
    n = int(math.sqrt(n_tok))
    grid = pos[1:].reshape(1, n, n, -1).permute(0, 3, 1, 2)
    out = F.interpolate(grid, scale_factor=((w0 + 0.1) / math.sqrt(n_tok), (h0 + 0.1) / math.sqrt(n_tok)), mode="bicubic")
    assert out.shape[-2] == w0 and out.shape[-1] == h0                   # :941
    return torch.cat([pos[:1], out.permute(0, 2, 3, 1).reshape(w0 * h0, -1)], dim=0)


def vit_embed(pixels: torch.Tensor, w: Dict[str, torch.Tensor], patch: int, eps: float) -> torch.Tensor:
    """CLIPVisionEmbeddings + pre_layrnorm: conv(stride=patch, no bias) -> flatten row-major
    -> prepend class_embedding -> + position_embedding -> LayerNorm (SURVEY 8a', a2).  The position table is the
    one AuroraEncoder.forward installs for this input size first (aurora.py:892)."""
    f = pixels.shape[0]
    p = F.conv2d(pixels, w["patch_embedding.weight"], stride=patch)      # [f, D, gh, gw]
    p = p.flatten(2).transpose(1, 2)                                      # [f, gh*gw, D]
    cls = w["class_embedding"].expand(f, 1, -1)
    pos = interpolate_pos_encoding(w["position_embedding.weight"], pixels.shape[-2], pixels.shape[-1], patch)
    x = torch.cat([cls, p], dim=1) + pos[None, : p.shape[1] + 1]
    d = x.shape[-1]
    return F.layer_norm(x, (d,), w["pre_layrnorm.weight"], w["pre_layrnorm.bias"], eps)


def vit_attention(x, size, lw, heads: int, q: Q = None):
    """aurora.py:621-701.  Returns (attn_out [n,t,D], metric [n,t,hd])."""
    q = q or _id
    n, t, d = x.shape
    hd = d // heads
    scale = hd ** -0.5
    qs = q(F.linear(x, lw["q_proj.weight"], lw["q_proj.bias"])) * scale     # :634 (scale on biased q)
    ks = q(F.linear(x, lw["k_proj.weight"], lw["k_proj.bias"]))
    vs = q(F.linear(x, lw["v_proj.weight"], lw["v_proj.bias"]))
    ks = ks.view(n, t, heads, hd).transpose(1, 2)
    vs = vs.view(n, t, heads, hd).transpose(1, 2)
    qs = qs.view(n, t, heads, hd).transpose(1, 2)
    metric = ks.mean(dim=1)                                                # :639
    aw = qs @ ks.transpose(-1, -2)                                         # [n,h,t,t]
    if size is not None:
        # :671-674 - size.log() is [n,t,1]: a per-query-row constant (softmax no-op)
        aw = aw + size.log()[:, None]
    aw = F.softmax(aw, dim=-1)
    o = (aw @ vs).transpose(1, 2).reshape(n, t, d)
    o = F.linear(q(o), lw["out_proj.weight"], lw["out_proj.bias"])
    return o, metric


def vit_layer(x, size, lw, heads, r, act, q: Q = None, forced_match=None, capture=None):
    """aurora.py:713-759 (LayerNorm eps = torch default 1e-5, :709,711)."""
    q = q or _id
    d = x.shape[-1]
    h = q(F.layer_norm(x, (d,), lw["layer_norm1.weight"], lw["layer_norm1.bias"], 1e-5))
    a, metric = vit_attention(h, size, lw, heads, q)
    x = q(x + a)
    match = bipartite_match(metric, r) if forced_match is None else forced_match
    if capture is not None:
        capture.append(dict(metric=metric, x_pre=x, size_pre=size, match=match))
    x, size = merge_wavg(x, size, match)
    x = q(x)
    h = q(F.layer_norm(x, (d,), lw["layer_norm2.weight"], lw["layer_norm2.bias"], 1e-5))
    h = q(ACT[act](F.linear(h, lw["fc1.weight"], lw["fc1.bias"])))
    h = F.linear(h, lw["fc2.weight"], lw["fc2.bias"])
    return q(x + h), size


def vit_encoder(x, layers, heads, r, act, q: Q = None, num_run: Optional[int] = None, capture=None):
    """aurora.py:772-860: returns the hidden_states tuple (state before each layer + final)."""
    size = None                                                            # :811
    states = []
    run = len(layers) if num_run is None else num_run
    for lw in layers[:run]:
        states.append(x)
        x, size = vit_layer(x, size, lw, heads, r, act, q, capture=capture)
    states.append(x)
    return states


def vit_features(pixels, vw, cfg, token_kept_ratio, q: Q = None, capture=None):
    """AuroraEncoder.forward (aurora.py:883-904) + select hidden_states[-2][:, 1:] (aurora.py:253).

    Only layers 0..L-2 are evaluated: the last layer and post_layernorm do not reach the output.
    """
    q = q or _id
    r = tome_r(pixels.shape[-2], pixels.shape[-1], cfg["patch_size"], token_kept_ratio,
               cfg["num_hidden_layers"])
    x = q(vit_embed(pixels, vw, cfg["patch_size"], cfg.get("layer_norm_eps", 1e-5)))
    L = cfg["num_hidden_layers"]
    states = vit_encoder(x, vw["layers"], cfg["num_attention_heads"], r, cfg["hidden_act"], q,
                         num_run=L - 1, capture=capture)
    return states[L - 1][:, 1:]


# --------------------------------------------------------------------------
# projector + splice  (modeling_projector.py:20-51, model/utils.py:138-295)
# --------------------------------------------------------------------------
_ACT = {"gelu": F.gelu, "quick_gelu": lambda x: x * torch.sigmoid(1.702 * x), "silu": F.silu, "swish": F.silu, "relu": F.relu,
        "gelu_new": lambda x: F.gelu(x, approximate="tanh"), "gelu_pytorch_tanh": lambda x: F.gelu(x, approximate="tanh")}


def projector(x, pw, q: Q = None, hidden_act: str = "gelu"):
    """modeling_projector.py:20-33: `model.0`, then (`ACT2FN[hidden_act]`, Linear `model.2k`) x (depth - 1), depth read off the keys;
    `bias=False` checkpoints simply have no bias keys (configuration_projector.py:9-22)"""
    q = q or _id
    depth = sum(1 for k in pw if k.endswith(".weight"))
    h = x
    for i in range(depth):
        h = F.linear(h, pw[f"model.{2 * i}.weight"], pw.get(f"model.{2 * i}.bias"))
        h = q(_ACT[hidden_act](h)) if i < depth - 1 else q(h)
    return h


def splice(input_ids: torch.Tensor, embed_table: torch.Tensor, visual: torch.Tensor) -> torch.Tensor:
    """prepare_inputs_labels_for_multimodal for batch 1: ids [n] (with -200 markers), visual
    [frames, n_kept, d] -> inputs_embeds [L0, d].  Image k replaces the k-th marker; markers
    beyond the number of frames are dropped (utils.py:228-233 'except: continue')."""
    out = []
    k = 0
    for tid in input_ids.tolist():
        if tid == IMAGE_TOKEN_INDEX:
            if k < visual.shape[0]:
                out.append(visual[k])
            k += 1
        else:
            out.append(embed_table[tid][None])
    return torch.cat(out, dim=0)


def splice_slowfast(input_ids: torch.Tensor, embed_table: torch.Tensor, visual: List[torch.Tensor]) -> torch.Tensor:
    """prepare_inputs_labels_for_multimodal_slowfast (model/utils.py:297-431) for batch 1: marker i takes
    visual[i] ([n_i, d], token counts differ per frame); unlike the plain splice a marker without a frame is an
    error (:361 indexes the list directly)."""
    out = []
    k = 0
    for tid in input_ids.tolist():
        if tid == IMAGE_TOKEN_INDEX:
            out.append(visual[k].reshape(-1, visual[k].shape[-1]))
            k += 1
        else:
            out.append(embed_table[tid][None])
    return torch.cat(out, dim=0)


def visual_features_slowfast(pixels, vw, pw, cfg, token_kept_ratio, q: Q = None, hidden_act: str = "gelu") -> List[torch.Tensor]:
    """aurora.py:223-246: frames 1.. at the current ratio, then frame 0 at ratio 1.0 (unmerged); each projected.
    Returns the list [frame0 [n0, d], frame1 [n, d], ...] that the slow-fast splice consumes."""
    low = vit_features(pixels[1:], vw, cfg, token_kept_ratio, q)
    high = vit_features(pixels[:1], vw, cfg, 1.0, q)
    return [projector(high[0], pw, q, hidden_act)] + [projector(x, pw, q, hidden_act) for x in low]


def build_prompt(prompt: str, num_images: int) -> str:
    """inference.py:76-85."""
    image_tokens = " ".join([DEFAULT_IMAGE_TOKEN] * num_images)
    return VICUNA_INSTRUCTION.format(input=image_tokens + "\n" + prompt, round=1)


def process_text(text: str, encode: Callable[[str, bool], List[int]]) -> List[int]:
    """inference.py:12-27.  encode(chunk, add_special_tokens)."""
    chunks = [encode(c, i == 0) for i, c in enumerate(text.split(DEFAULT_IMAGE_TOKEN))]
    ids: List[int] = []
    for i, c in enumerate(chunks):
        ids.extend(c)
        if i != len(chunks) - 1:
            ids.append(IMAGE_TOKEN_INDEX)
    return ids


# --------------------------------------------------------------------------
# Llama greedy decode (third-party transformers semantics; SURVEY 8a' "Llama")
# --------------------------------------------------------------------------
def rmsnorm(x, w, eps):
    v = x.pow(2).mean(-1, keepdim=True)
    return w * (x * torch.rsqrt(v + eps))


def rope_cos_sin(positions: torch.Tensor, head_dim: int, theta: float, factor: float):
    inv = 1.0 / (theta ** (torch.arange(0, head_dim, 2, dtype=torch.float32) / head_dim))
    ang = (positions.to(torch.float32) / factor)[:, None] * inv[None, :]
    emb = torch.cat([ang, ang], dim=-1)
    return emb.cos(), emb.sin()


def _rot_half(x):
    h = x.shape[-1] // 2
    return torch.cat([-x[..., h:], x[..., :h]], dim=-1)


def llama_forward(x, lw, cfg, kv: Optional[list], pos0: int, q: Q = None):
    """x [T, d] new positions pos0..pos0+T-1 for ONE sequence.  kv: list per layer of
    (K [h, ctx, hd], V [h, ctx, hd]) or None.  Returns (hidden [T, d] after final norm, new kv)."""
    q = q or _id
    T, d = x.shape
    H, hd = cfg["num_attention_heads"], cfg["hidden_size"] // cfg["num_attention_heads"]
    eps = cfg["rms_norm_eps"]
    cos, sin = rope_cos_sin(torch.arange(pos0, pos0 + T), hd, cfg["rope_theta"], cfg.get("rope_factor", 1.0))
    new_kv = []
    for li, l in enumerate(lw["layers"]):
        h = q(rmsnorm(x, l["input_layernorm.weight"], eps))
        qq = F.linear(h, l["q_proj.weight"]).view(T, H, hd).transpose(0, 1)
        kk = F.linear(h, l["k_proj.weight"]).view(T, H, hd).transpose(0, 1)
        vv = q(F.linear(h, l["v_proj.weight"])).view(T, H, hd).transpose(0, 1)
        qq = q(qq * cos + _rot_half(qq) * sin)
        kk = q(kk * cos + _rot_half(kk) * sin)
        if kv is not None:
            kk = torch.cat([kv[li][0], kk], dim=1)
            vv = torch.cat([kv[li][1], vv], dim=1)
        new_kv.append((kk, vv))
        ctx = kk.shape[1]
        s = (qq @ kk.transpose(-1, -2)) / math.sqrt(hd)
        mask = torch.arange(ctx)[None, :] > (torch.arange(T)[:, None] + pos0)
        s = s.masked_fill(mask[None], -math.inf)
        o = (F.softmax(s, dim=-1) @ vv).transpose(0, 1).reshape(T, d)
        x = q(x + F.linear(q(o), l["o_proj.weight"]))
        h = q(rmsnorm(x, l["post_attention_layernorm.weight"], eps))
        g = F.linear(h, l["gate_proj.weight"])
        u = F.linear(h, l["up_proj.weight"])
        x = q(x + F.linear(q(F.silu(g) * u), l["down_proj.weight"]))
    return rmsnorm(x, lw["norm.weight"], eps), new_kv


def llama_greedy(embeds, lw, cfg, max_new_tokens: int, eos_id: Optional[int] = 2, q: Q = None,
                 return_logits: bool = False):
    """generate(inputs_embeds=..., do_sample=False) (inference.py:89-96): prefill, then one
    position per step with a KV cache, argmax, stop at EOS or max_new_tokens; returns new ids only."""
    q = q or _id
    h, kv = llama_forward(embeds, lw, cfg, None, 0, q)
    ids, logits_all = [], []
    pos = embeds.shape[0]
    for _ in range(max_new_tokens):
        logits = F.linear(q(h[-1:]), lw["lm_head.weight"]).float()
        logits_all.append(logits[0])
        nxt = int(torch.argmax(logits[0]))
        ids.append(nxt)
        if eos_id is not None and nxt == eos_id:
            break
        if len(ids) == max_new_tokens:
            break
        h, kv = llama_forward(lw["embed_tokens.weight"][nxt][None], lw, cfg, kv, pos, q)
        pos += 1
    if return_logits:
        return ids, torch.stack(logits_all)
    return ids


# --------------------------------------------------------------------------
# whole path: AuroraModel.forward(mode="inference") + generate  (aurora.py:214-270)
# --------------------------------------------------------------------------
def caption_ids(pixels, input_ids, weights, cfg, token_kept_ratio, max_new_tokens,
                eos_id: Optional[int] = 2, q: Q = None, timings: Optional[dict] = None, return_logits: bool = False):
    """pixels [f,3,H,W] fp32 (already normalised), input_ids list[int] with -200 markers."""
    import time
    t0 = time.perf_counter()
    feats = vit_features(pixels, weights["vit"], cfg["vit"], token_kept_ratio, q)    # [f, n, Dv]
    f, n, dv = feats.shape
    vis = projector(feats.reshape(1, f * n, dv), weights["projector"], q, (cfg.get("projector") or {}).get("hidden_act", "gelu")).reshape(f, n, -1)
    emb = splice(torch.tensor(input_ids), weights["llm"]["embed_tokens.weight"], vis)
    t1 = time.perf_counter()
    ids, logits = llama_greedy(emb, weights["llm"], cfg["llm"], max_new_tokens, eos_id, q, return_logits=True)
    t2 = time.perf_counter()
    if timings is not None:
        timings.update(vision_s=t1 - t0, llm_s=t2 - t1, prefill_len=emb.shape[0])
    return (ids, logits) if return_logits else ids
