"""Host side of the MI355X AuroraCap path: owns torch-allocated device memory (weights, workspace, KV
pool), re-lays weights into the kernels' fragment format once, and drives libaurora_hip.so through its
C ABI.  PyTorch is used for memory / stream housekeeping only - no torch compute op is on the hot path.

Weight dictionaries use the reference checkpoints' parameter names (HF CLIP vision tower, xtuner
projector `model.0 / model.2`, HF Llama), grouped as
    {"vit": {...,"layers":[{...}]}, "projector": {...}, "llm": {...,"layers":[{...}]}}
exactly like `oracle/aurora_oracle.py` consumes them, so parity tests feed both sides the same tensors.
"""
from __future__ import annotations

import ctypes as C
from typing import Dict, List, Optional, Sequence, Tuple

import numpy as np
import torch

from . import _lib
from ._lib import AurConfig, check

IMAGE_TOKEN_INDEX = -200     # src/xtuner/xtuner/utils/constants.py:4


def _rup(x: int, m: int) -> int:
    return (x + m - 1) // m * m


def tome_r(height: int, width: int, patch: int, token_kept_ratio: float, num_layers: int) -> int:
    """aurora.py:895 (evaluated inside the C library in doubles, same operation order)."""
    return int(_lib.lib().aur_tome_r(height, width, patch, float(token_kept_ratio), num_layers))


def tokens_at_layer(t0: int, r: int, layer: int) -> int:
    return int(_lib.lib().aur_tokens_at_layer(t0, r, layer))


def projector_settings(pc: dict, visual_hidden: Optional[int], llm_hidden: Optional[int]) -> dict:
    """Validated ProjectorConfig values (configuration_projector.py:9-22; defaults = the class's own: depth 2, 'gelu', bias True).
    Raises on what the kernels do not implement instead of loading it wrong."""
    depth, act, bias = int(pc.get("depth", 2)), pc.get("hidden_act", "gelu"), bool(pc.get("bias", True))
    if not 1 <= depth <= 8:
        raise NotImplementedError(f"projector depth {depth}: 1 .. 8 Linear layers are supported")
    if act not in _lib.ACT_BY_NAME:
        raise NotImplementedError(f"projector hidden_act {act!r}: the GEMM epilogues implement {sorted(_lib.ACT_BY_NAME)}")
    for key, have in (("visual_hidden_size", visual_hidden), ("llm_hidden_size", llm_hidden)):
        if key in pc and have is not None and int(pc[key]) != int(have):
            raise ValueError(f"projector config {key} = {pc[key]} but the {'vision tower' if key[0] == 'v' else 'language model'} has hidden size {have}")
    return dict(depth=depth, hidden_act=act, bias=bias)


class AuroraCapEngine:
    def __init__(self, cfg: dict, weights: dict, *, max_frames: int = 8, max_batch: int = 1, max_ctx: int = 4096,
                 max_new_tokens: int = 256, page_tokens: int = 64, use_graph: bool = True, num_banks: int = 1,
                 max_image: Optional[int] = None, device: str = "cuda:0", spare_slots: int = 0):
        if not torch.cuda.is_available():
            raise _lib.AuroraHipError("AuroraCapEngine needs a ROCm GPU (torch.cuda.is_available() is False); "
                                      "there is no CPU fallback")
        self.L = _lib.lib()
        self.dev = self.normalize_device(device)
        torch.cuda.set_device(self.dev)
        # All work is enqueued on ONE explicit (non-null) HIP stream: the null stream cannot be captured into
        # a hipGraph.  It becomes this thread's current torch stream, so caller-side tensor housekeeping is
        # ordered with the kernels without cross-stream events.
        self._prev_stream = torch.cuda.current_stream(self.dev)
        self.stream = torch.cuda.Stream(self.dev)
        self.stream.wait_stream(self._prev_stream)
        torch.cuda.set_stream(self.stream)
        self.cfg = cfg
        v, l = cfg.get("vit"), cfg.get("llm")
        c = AurConfig()
        # unused sub-models get harmless valid dims so that aur_create's checks pass
        vv = v or dict(hidden_size=64, num_attention_heads=4, num_hidden_layers=2, intermediate_size=64, patch_size=14,
                       image_size=28, hidden_act="quick_gelu")
        ll = l or dict(hidden_size=128, num_attention_heads=4, num_hidden_layers=1, intermediate_size=128, vocab_size=128,
                       rms_norm_eps=1e-5, rope_theta=1e4)
        c.vit_hidden, c.vit_heads, c.vit_layers = vv["hidden_size"], vv["num_attention_heads"], vv["num_hidden_layers"]
        # vit_image is the CAPACITY (largest input side the workspace admits); the checkpoint's position table belongs to
        # image_size.  max_image > image_size admits larger inputs through the interpolated table (aurora.py:909-951).
        self.max_image = max(int(max_image or 0), vv["image_size"]) // vv["patch_size"] * vv["patch_size"]
        c.vit_mlp, c.vit_patch, c.vit_image = vv["intermediate_size"], vv["patch_size"], self.max_image
        c.vit_native_image = vv["image_size"]
        c.vit_channels = vv.get("num_channels", 3)
        act = vv.get("hidden_act", "quick_gelu")
        if act not in ("quick_gelu", "gelu"):
            raise ValueError(f"unsupported ViT hidden_act {act!r}")
        c.vit_act = _lib.AUR_ACT_GELU if act == "gelu" else _lib.AUR_ACT_QUICK_GELU
        c.vit_ln_eps = vv.get("layer_norm_eps", 1e-5)
        c.llm_hidden, c.llm_heads, c.llm_layers = ll["hidden_size"], ll["num_attention_heads"], ll["num_hidden_layers"]
        c.llm_mlp, c.llm_vocab = ll["intermediate_size"], ll["vocab_size"]
        c.llm_rms_eps, c.rope_theta, c.rope_factor = ll["rms_norm_eps"], ll["rope_theta"], ll.get("rope_factor", 1.0)
        c.max_frames, c.max_batch, c.max_ctx = max_frames, max_batch, max_ctx
        c.max_new_tokens, c.page_tokens, c.use_graph = max_new_tokens, page_tokens, int(use_graph)
        c.num_banks = num_banks
        c.spare_slots = spare_slots
        # projector (configuration_projector.py:9-22): depth / hidden_act / bias come from projector/config.json (cfg["projector"],
        # aurora_amd.checkpoint.projector_config); without one: the AuroraCap defaults, whose depth is read off the weight keys
        pc = dict(cfg.get("projector") or {})
        if "depth" not in pc and "projector" in weights:
            pc["depth"] = sum(1 for k in weights["projector"] if k.endswith(".weight"))
        self.proj_cfg = projector_settings(pc, vv["hidden_size"] if v else None, ll["hidden_size"] if l else None)
        c.proj_depth, c.proj_act = self.proj_cfg["depth"], _lib.ACT_BY_NAME[self.proj_cfg["hidden_act"]]
        self.spare_slots = spare_slots
        self._front_graphs = {}                          # shape bucket -> FrontEndGraph (caption_stream's captured front ends), LRU order
        self.front_graph_cap = 16
        self._bank_state = {0: (0, 0), 1: (0, 0)}        # bank -> (batch, max_new) for outputs()
        self._bank = 0
        self.c = c
        self.v, self.l = v, l
        self.max_new_tokens = max_new_tokens
        self.max_batch = max_batch
        self.ctx = C.c_void_p()
        rc = self.L.aur_create(C.byref(c), C.byref(self.ctx))
        if rc != 0:
            raise _lib.AuroraHipError(f"aur_create failed ({rc}): {self.L.aur_last_error(None).decode()}")
        self._keep: List[torch.Tensor] = []
        self.ws = torch.empty(self.L.aur_workspace_bytes(self.ctx), dtype=torch.uint8, device=self.dev)
        check(self.ctx, self.L.aur_set_workspace(self.ctx, self.ws.data_ptr(), self.ws.numel()), "aur_set_workspace")
        self.kv = None
        if l is not None:
            self.kv = torch.empty(self.L.aur_kv_pool_bytes(self.ctx), dtype=torch.uint8, device=self.dev)
            check(self.ctx, self.L.aur_set_kv_pool(self.ctx, self.kv.data_ptr(), self.kv.numel()), "aur_set_kv_pool")
        if v is not None and "vit" in weights:
            self._load_vit(weights["vit"])
        if l is not None and "llm" in weights:
            self._load_llm(weights["llm"])
            if "projector" in weights:
                self._load_projector(weights["projector"])
        if (v is not None and "vit" in weights) or (l is not None and "llm" in weights):
            check(self.ctx, self.L.aur_finalize(self.ctx, self._stream()), "aur_finalize")
        torch.cuda.synchronize()

    # ------------------------------------------------------------------ housekeeping
    @staticmethod
    def normalize_device(device) -> torch.device:
        """"cuda" (no ordinal, the lmms-eval adaptor's default) -> the process's current GPU: torch.cuda.set_device and the
        allocations below need an indexed device."""
        dev = torch.device(device)
        if dev.type != "cuda":
            raise _lib.AuroraHipError(f"AuroraCapEngine runs on a ROCm GPU only (device={device!r}); there is no CPU fallback")
        if dev.index is None:
            dev = torch.device("cuda", torch.cuda.current_device())
        return dev

    def close(self):
        if getattr(self, "ctx", None):
            torch.cuda.synchronize()
            self._front_graphs = {}                                 # the graphs themselves die with the ctx
            self.L.aur_destroy(self.ctx)
            self.ctx = None
            self._masked = {}                                       # CU-masked streams are shared per process (streams.py): never destroyed
            if torch.cuda.current_stream(self.dev) == self.stream:
                torch.cuda.set_stream(self._prev_stream)

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass

    def _stream(self):
        cur = torch.cuda.current_stream(self.dev)
        if cur.cuda_stream != 0:            # any explicit stream the caller made current (e.g. a second pipeline stream)
            return C.c_void_p(cur.cuda_stream)
        self.stream.wait_stream(cur)        # null stream: order ours after it (it cannot be graph-captured)
        return C.c_void_p(self.stream.cuda_stream)

    def _h(self, t: torch.Tensor) -> torch.Tensor:
        return t.detach().to(device=self.dev, dtype=torch.float16).contiguous()

    def _f(self, t: torch.Tensor) -> torch.Tensor:
        return t.detach().to(device=self.dev, dtype=torch.float32).contiguous()

    def _set(self, name: str, t: torch.Tensor):
        self._keep.append(t)
        check(self.ctx, self.L.aur_set_tensor(self.ctx, name.encode(), t.data_ptr(), t.numel() * t.element_size()), name)

    def pack(self, w: torch.Tensor, npad: int, kpad: int, row_map: Optional[np.ndarray] = None) -> torch.Tensor:
        """row-major [N, K] -> MFMA fragment tiles (aur_pack_linear)."""
        w = self._h(w)
        n, k = w.shape
        out = torch.empty(npad * kpad, dtype=torch.float16, device=self.dev)
        rm = None
        if row_map is not None:
            assert row_map.shape[0] == npad
            rm = torch.from_numpy(row_map.astype(np.int32)).to(self.dev)
        check(self.ctx, self.L.aur_pack_linear(self.ctx, w.data_ptr(), n, k, k, rm.data_ptr() if rm is not None else None,
                                               npad, kpad, out.data_ptr(), self._stream()), "aur_pack_linear")
        torch.cuda.current_stream().synchronize()
        return out

    def _bias(self, b: Optional[torch.Tensor], npad: int, row_map: Optional[np.ndarray] = None) -> torch.Tensor:
        out = torch.zeros(npad, dtype=torch.float32, device=self.dev)
        if b is None:
            return out
        b = self._f(b)
        if row_map is None:
            out[: b.numel()] = b
        else:
            rm = torch.from_numpy(row_map.astype(np.int64)).to(self.dev)
            ok = rm >= 0
            out[ok] = b[rm[ok]]
        return out

    # ------------------------------------------------------------------ weight loading
    def _load_vit(self, w: dict):
        v = self.v
        D, H, mlp = v["hidden_size"], v["num_attention_heads"], v["intermediate_size"]
        hd = D // H
        hdp = _rup(hd, 32)
        qcols = _rup(H * hdp, 64)
        npad = _rup(2 * qcols + D, 256)
        dpad, mpad = _rup(D, 256), _rup(mlp, 256)
        P, Cn = v["patch_size"], v.get("num_channels", 3)
        kpad = _rup(Cn * P * P, 64)
        # fused QKV row map: Q and K heads padded hd -> hd_pad (zero rows), V natural
        rm = np.full(npad, -1, np.int32)
        for h in range(H):
            rm[h * hdp: h * hdp + hd] = np.arange(h * hd, (h + 1) * hd)
            rm[qcols + h * hdp: qcols + h * hdp + hd] = D + np.arange(h * hd, (h + 1) * hd)
        rm[2 * qcols: 2 * qcols + D] = 2 * D + np.arange(D)
        self._set("vit.patch.w", self.pack(w["patch_embedding.weight"].reshape(D, -1), dpad, kpad))
        self._set("vit.cls", self._h(w["class_embedding"].reshape(-1)))
        self._pos_cache: Dict[tuple, torch.Tensor] = {}                  # interpolated_pos(): one table per input size
        self._set("vit.pos", self._h(w["position_embedding.weight"]))
        self._set("vit.preln.w", self._f(w["pre_layrnorm.weight"]))
        self._set("vit.preln.b", self._f(w["pre_layrnorm.bias"]))
        nl = v["num_hidden_layers"] - 1          # hidden_states[-2]: the last layer is never needed
        for i in range(nl):
            lw = w["layers"][i]
            p = f"vit.{i}."
            for a in ("ln1", "ln2"):
                full = "layer_norm1" if a == "ln1" else "layer_norm2"
                self._set(p + a + ".w", self._f(lw[full + ".weight"]))
                self._set(p + a + ".b", self._f(lw[full + ".bias"]))
            wqkv = torch.cat([lw["q_proj.weight"], lw["k_proj.weight"], lw["v_proj.weight"]], 0)
            bqkv = torch.cat([lw["q_proj.bias"], lw["k_proj.bias"], lw["v_proj.bias"]], 0)
            self._set(p + "qkv.w", self.pack(wqkv, npad, D, rm))
            self._set(p + "qkv.b", self._bias(bqkv, npad, rm))
            self._set(p + "out.w", self.pack(lw["out_proj.weight"], dpad, D))
            self._set(p + "out.b", self._bias(lw["out_proj.bias"], dpad))
            self._set(p + "fc1.w", self.pack(lw["fc1.weight"], mpad, D))
            self._set(p + "fc1.b", self._bias(lw["fc1.bias"], mpad))
            self._set(p + "fc2.w", self.pack(lw["fc2.weight"], dpad, mlp))
            self._set(p + "fc2.b", self._bias(lw["fc2.bias"], dpad))

    @staticmethod
    def llama_qkv_row_map(d: int, heads: int, npad: int) -> np.ndarray:
        """Q/K rows permuted per head so that fragment tile 2p holds d in [16p, 16p+16) and tile 2p+1 its RoPE
        partner d + hd/2: the rotation (HF rotate_half) happens in-lane in the GEMM epilogue."""
        hd = d // heads
        rm = np.full(npad, -1, np.int32)
        per_head = np.empty(hd, np.int64)
        for p in range(hd // 32):
            per_head[32 * p: 32 * p + 16] = 16 * p + np.arange(16)
            per_head[32 * p + 16: 32 * p + 32] = hd // 2 + 16 * p + np.arange(16)
        for h in range(heads):
            rm[h * hd: (h + 1) * hd] = h * hd + per_head
            rm[d + h * hd: d + (h + 1) * hd] = d + h * hd + per_head
        rm[2 * d: 3 * d] = 2 * d + np.arange(d)
        return rm

    def _load_llm(self, w: dict):
        l = self.l
        d, H, mlp, V = l["hidden_size"], l["num_attention_heads"], l["intermediate_size"], l["vocab_size"]
        qkv_npad, gu_npad, dpad, vpad = _rup(3 * d, 256), _rup(2 * mlp, 256), _rup(d, 256), _rup(V, 256)
        rm_qkv = self.llama_qkv_row_map(d, H, qkv_npad)
        rm_gu = np.full(gu_npad, -1, np.int32)
        rm_gu[0: 2 * mlp: 2] = np.arange(mlp)
        rm_gu[1: 2 * mlp: 2] = mlp + np.arange(mlp)
        # RMSNorm weights are FOLDED into the projection that consumes the normalised activations:
        #   W (w_norm * x_hat) = (W diag(w_norm)) x_hat.  The kernels then only need the per-row 1/rms
        #   (prefill: weight-less norm kernel; decode: sum(x^2) carried by the residual-producing epilogues).
        def fold(wmat, wnorm):
            return (wmat.detach().to(self.dev, torch.float32) * wnorm.detach().to(self.dev, torch.float32)[None, :]).to(torch.float16)

        self._set("llm.embed", self._h(w["embed_tokens.weight"]))
        self._set("llm.lm_head.w", self.pack(fold(w["lm_head.weight"], w["norm.weight"]), vpad, d))
        for i, lw in enumerate(w["layers"]):
            p = f"llm.{i}."
            wqkv = torch.cat([lw["q_proj.weight"], lw["k_proj.weight"], lw["v_proj.weight"]], 0)
            self._set(p + "qkv.w", self.pack(fold(wqkv, lw["input_layernorm.weight"]), qkv_npad, d, rm_qkv))
            self._set(p + "o.w", self.pack(lw["o_proj.weight"], dpad, d))
            wgu = torch.cat([lw["gate_proj.weight"], lw["up_proj.weight"]], 0)
            self._set(p + "gateup.w", self.pack(fold(wgu, lw["post_attention_layernorm.weight"]), gu_npad, d, rm_gu))
            self._set(p + "down.w", self.pack(lw["down_proj.weight"], dpad, mlp))

    def _load_projector(self, w: dict):
        """modeling_projector.py:20-33: `model.0`, then (`ACT2FN[hidden_act]`, `model.2k`) for k = 1 .. depth - 1.  A bias-free projector
        (`bias=False`) loads zeros.  Keys, shapes and count are checked against the config: nothing loads silently wrong."""
        d, dv, pc = self.l["hidden_size"], self.v["hidden_size"], self.proj_cfg
        dpad = _rup(d, 256)
        want = {f"model.{2 * i}.weight" for i in range(pc["depth"])}
        have = {k for k in w if k.endswith(".weight")}
        if want != have:
            raise ValueError(f"projector weights {sorted(have)} do not match depth {pc['depth']} (expected {sorted(want)})")
        for i in range(pc["depth"]):
            wt, k_in = w[f"model.{2 * i}.weight"], (dv if i == 0 else d)
            if tuple(wt.shape) != (d, k_in):
                raise ValueError(f"projector model.{2 * i}.weight has shape {tuple(wt.shape)}, the configs say {(d, k_in)}")
            b = w.get(f"model.{2 * i}.bias")
            if (b is not None) != pc["bias"]:
                raise ValueError(f"projector model.{2 * i}.bias {'present' if b is not None else 'missing'} but config.bias is {pc['bias']}")
            self._set(f"proj.{i}.w", self.pack(wt, dpad, k_in))
            self._set(f"proj.{i}.b", self._bias(b if b is not None else torch.zeros(d), dpad))

    # ------------------------------------------------------------------ hot path
    def tome_r(self, token_kept_ratio: float, height: Optional[int] = None, width: Optional[int] = None) -> int:
        v = self.v
        return tome_r(height or v["image_size"], width or v["image_size"], v["patch_size"], token_kept_ratio,
                      v["num_hidden_layers"])

    def vit_encode(self, pixels: torch.Tensor, r: int, out: Optional[torch.Tensor] = None) -> torch.Tensor:
        """pixels [F, C, H, W] -> hidden_states[-2][:, 1:]  as fp16 [F, n_kept, D] (into `out` when given: a captured front
        end must not allocate)."""
        v = self.v
        if pixels.dim() != 4 or pixels.shape[1] != v.get("num_channels", 3):
            raise ValueError(f"pixel_values must be [frames, {v.get('num_channels', 3)}, H, W], got {tuple(pixels.shape)}")
        px = self._h(pixels)
        F, H, W = px.shape[0], px.shape[-2], px.shape[-1]
        t0 = (H // v["patch_size"]) * (W // v["patch_size"]) + 1
        if t0 < 2 or t0 > (self.max_image // v["patch_size"]) ** 2 + 1:
            raise ValueError(f"a {H}x{W} input has {t0} tokens per frame; this engine holds up to "
                             f"{(self.max_image // v['patch_size']) ** 2 + 1} (max_image {self.max_image})")
        n_kept = tokens_at_layer(t0, r, v["num_hidden_layers"] - 1) - 1
        if out is None:
            out = torch.empty(F, n_kept, v["hidden_size"], dtype=torch.float16, device=self.dev)
        else:
            assert out.shape == (F, n_kept, v["hidden_size"]) and out.dtype == torch.float16 and out.is_contiguous() and out.is_cuda
        nk = C.c_int32(0)
        pos = self.interpolated_pos(H, W)
        check(self.ctx, self.L.aur_vit_encode_hw(self.ctx, px.data_ptr(), F, H, W, pos.data_ptr() if pos is not None else None, r,
                                                 out.data_ptr(), C.byref(nk), self._stream()), "aur_vit_encode")
        assert nk.value == n_kept
        return out

    def interpolated_pos(self, height: int, width: int) -> Optional[torch.Tensor]:
        """Position table for a non-native input size (aurora.py:909-951): the n x n patch rows of the checkpoint's table
        resampled bicubically to (height // patch, width // patch) with the reference's scale factors (g + 0.1) / n; None
        for the native grid.  One launch per input size (`pos_interp_kernel`, vit.hip), cached."""
        v = self.v
        gh, gw = height // v["patch_size"], width // v["patch_size"]
        n = v["image_size"] // v["patch_size"]
        if gh == n and gw == n:
            return None
        key = (gh, gw)
        if key not in self._pos_cache:
            tab = torch.empty(1 + gh * gw, v["hidden_size"], dtype=torch.float16, device=self.dev)
            check(self.ctx, self.L.aur_vit_pos_interp(self.ctx, height, width, tab.data_ptr(), self._stream()), "aur_vit_pos_interp")
            self._pos_cache[key] = tab
        return self._pos_cache[key]

    def splice_plan(self, input_ids: Sequence[int], frames: int, n_kept, strict: bool = False):
        """Destination-row maps of the prefix splice for one prompt (model/utils.py:198-240 semantics: marker k takes
        frame k; markers beyond the number of frames are dropped), uploaded once.  Re-usable across calls: building it
        involves blocking host->device copies, which must stay out of a multi-stream pipeline's steady state."""
        counts = [int(n_kept)] * frames if np.isscalar(n_kept) else [int(c) for c in n_kept]   # per-frame rows (slow-fast: ragged)
        vis_rows, text_ids, text_rows = [], [], []
        row, k = 0, 0
        for tid in input_ids:
            if tid == IMAGE_TOKEN_INDEX:
                if k < frames:
                    vis_rows.append(np.arange(row, row + counts[k]))
                    row += counts[k]
                elif strict:
                    raise IndexError(f"image marker {k} has no frame ({frames} frames)")      # model/utils.py:361
                k += 1
            else:
                text_ids.append(int(tid))
                text_rows.append(row)
                row += 1
        vr = np.concatenate(vis_rows) if vis_rows else np.zeros(0, np.int64)
        return dict(seq_len=row, used=len(vis_rows), n_kept=n_kept, nvis=int(sum(counts[:len(vis_rows)])), ntext=len(text_ids),
                    vis_rows=torch.from_numpy(vr.astype(np.int32)).to(self.dev),
                    text_ids=torch.tensor(text_ids, dtype=torch.int32, device=self.dev),
                    text_rows=torch.tensor(text_rows, dtype=torch.int32, device=self.dev))

    def project_splice(self, vis: torch.Tensor, input_ids: Optional[Sequence[int]] = None, out: Optional[torch.Tensor] = None,
                       plan: Optional[dict] = None):
        """vis [frames, n_kept, Dv] + ids (with -200 markers) -> (embeds [L_pad, d] fp16, seq_len)."""
        d = self.l["hidden_size"]
        if vis.dim() == 2:                    # flat rows in marker order (frames with different token counts: slow-fast)
            assert plan is not None and plan["nvis"] <= vis.shape[0]
            seq_len = plan["seq_len"]
            vflat = self._h(vis[:plan["nvis"]])
        else:
            if plan is None:
                plan = self.splice_plan(input_ids, vis.shape[0], vis.shape[1])
            assert plan["n_kept"] == vis.shape[1] and plan["used"] <= vis.shape[0]
            seq_len, used = plan["seq_len"], plan["used"]
            vflat = self._h(vis[:used]).reshape(used * plan["n_kept"], -1)
        embeds = out if out is not None else torch.empty(_rup(seq_len, 32), d, dtype=torch.float16, device=self.dev)
        assert embeds.shape[0] >= _rup(seq_len, 32) and embeds.is_contiguous()
        check(self.ctx, self.L.aur_project_splice(self.ctx, vflat.data_ptr(), vflat.shape[0], plan["vis_rows"].data_ptr(),
                                                  plan["text_ids"].data_ptr(), plan["text_rows"].data_ptr(), plan["ntext"], seq_len,
                                                  embeds.data_ptr(), self._stream()), "aur_project_splice")
        self._tmp = getattr(self, "_tmp", [])[-64:] + [(vflat, plan)]      # keep alive until consumed
        return embeds, seq_len

    def select_bank(self, bank: int):
        """Switch the generation bank (double-buffered KV slots / state) that the following calls act on."""
        self._bank_state[self._bank] = (getattr(self, "_batch", 0), getattr(self, "_max_new", 0))
        check(self.ctx, self.L.aur_select_bank(self.ctx, bank), "aur_select_bank")
        self._bank = bank
        self._batch, self._max_new = self._bank_state[bank]

    def begin_batch(self, batch: int, max_new_tokens: int, eos_id: Optional[int]):
        self._batch, self._max_new = batch, max_new_tokens
        check(self.ctx, self.L.aur_begin_batch(self.ctx, batch, max_new_tokens, -1 if eos_id is None else int(eos_id),
                                               self._stream()), "aur_begin_batch")

    def prefill(self, slot: int, embeds: torch.Tensor, seq_len: int):
        check(self.ctx, self.L.aur_llm_prefill(self.ctx, slot, embeds.data_ptr(), seq_len, self._stream()), "aur_llm_prefill")

    def prefill_batch(self, slot0: int, nseq: int, embeds: torch.Tensor, seq_len: int):
        """embeds [nseq * round_up(seq_len, 32), d]: equal-length sequences prefetched in ONE pass (large-M GEMMs)."""
        check(self.ctx, self.L.aur_llm_prefill_batch(self.ctx, slot0, nseq, embeds.data_ptr(), seq_len, self._stream()),
              "aur_llm_prefill_batch")

    def prefill_stage(self, seq0: int, nseq: int, embeds: torch.Tensor, seq_len: int):
        """The layer stack of `prefill_batch` into KV sequences [seq0, seq0 + nseq) (normally the spare ones,
        `max_batch ..`): touches no decode state, so it may run on another stream while `decode` runs."""
        check(self.ctx, self.L.aur_llm_prefill_stage(self.ctx, seq0, nseq, embeds.data_ptr(), seq_len, self._stream()),
              "aur_llm_prefill_stage")

    def prefill_commit(self, slot0: int, nseq: int, seq0: int, embeds: torch.Tensor, seq_len: int):
        """On the decode stream, after `prefill_stage` has completed: slots [slot0 ..) take over the pages of sequences
        [seq0 ..) (which get the slots' old pages), are reset, and receive their first tokens."""
        check(self.ctx, self.L.aur_llm_prefill_commit(self.ctx, slot0, nseq, seq0, embeds.data_ptr(), seq_len, self._stream()),
              "aur_llm_prefill_commit")

    def decode(self, steps: int):
        check(self.ctx, self.L.aur_llm_decode(self.ctx, steps, self._stream()), "aur_llm_decode")

    def outputs(self):
        ids = np.zeros((self._batch, self._max_new), np.int32)
        lens = np.zeros(self._batch, np.int32)
        check(self.ctx, self.L.aur_get_outputs(self.ctx, ids.ctypes.data_as(C.POINTER(C.c_int32)),
                                               lens.ctypes.data_as(C.POINTER(C.c_int32)), self._stream()), "aur_get_outputs")
        return [ids[b, : lens[b]].tolist() for b in range(self._batch)]

    def unfinished(self) -> int:
        n = C.c_int32(0)
        check(self.ctx, self.L.aur_unfinished(self.ctx, C.byref(n), self._stream()), "aur_unfinished")
        return n.value

    def logits(self) -> torch.Tensor:
        """fp32 [batch, vocab] logits of the most recent prefill / decode step (copy)."""
        out = torch.empty(self._batch, self.l["vocab_size"], dtype=torch.float32, device=self.dev)
        check(self.ctx, self.L.aur_copy_logits(self.ctx, out.data_ptr(), self._stream()), "aur_copy_logits")
        torch.cuda.synchronize()
        return out

    def generate(self, embeds_list, seq_lens, max_new_tokens: int, eos_id: Optional[int] = 2, check_every: int = 32):
        """Greedy generation for a batch of already-spliced prefixes (one KV slot per sequence)."""
        B = len(embeds_list)
        self.begin_batch(B, max_new_tokens, eos_id)
        for b in range(B):
            self.prefill(b, embeds_list[b], seq_lens[b])
        done = 1
        while done < max_new_tokens:
            n = min(check_every, max_new_tokens - done) if eos_id is not None else max_new_tokens - done
            self.decode(n)
            done += n
            if eos_id is not None and self.unfinished() == 0:
                break
        return self.outputs()

    def caption_ids(self, pixels: torch.Tensor, input_ids: Sequence[int], token_kept_ratio: float, max_new_tokens: int,
                    eos_id: Optional[int] = 2) -> List[int]:
        """Whole path for one clip: the call sequence of inference.py:87-96."""
        r = self.tome_r(token_kept_ratio, pixels.shape[-2], pixels.shape[-1])
        vis = self.vit_encode(pixels, r)
        emb, L = self.project_splice(vis, input_ids)
        return self.generate([emb], [L], max_new_tokens, eos_id)[0]

    def caption_batch(self, clips: Sequence[Tuple[torch.Tensor, Sequence[int]]], token_kept_ratio: float, max_new_tokens: int,
                      eos_id: Optional[int] = 2, prefill_group: int = 8, check_every: int = 32) -> List[List[int]]:
        """Whole path for up to max_batch clips at once: (pixel_values [f, C, H, W], input_ids with -200 markers) each.
        Clips may differ in frame count and prompt; neighbours whose spliced length is equal share one prefill pass.
        Every kernel on the path is batch-invariant, so each clip gets exactly the ids it gets when captioned alone."""
        B = len(clips)
        if not 1 <= B <= self.max_batch:
            raise ValueError(f"{B} clips for an engine built with max_batch={self.max_batch}")
        d = self.l["hidden_size"]
        rs, plans = [], []
        for px, ids in clips:
            r = self.tome_r(token_kept_ratio, px.shape[-2], px.shape[-1])
            t0 = (px.shape[-2] // self.v["patch_size"]) * (px.shape[-1] // self.v["patch_size"]) + 1     # as vit_encode counts them
            rs.append(r)
            plans.append(self.splice_plan(ids, px.shape[0], tokens_at_layer(t0, r, self.v["num_hidden_layers"] - 1) - 1))
        self.begin_batch(B, max_new_tokens, eos_id)
        b = 0
        while b < B:
            L, g = plans[b]["seq_len"], 1
            while b + g < B and g < prefill_group and plans[b + g]["seq_len"] == L:
                g += 1
            Lp = _rup(L, 32)
            buf = torch.empty(g * Lp, d, dtype=torch.float16, device=self.dev)
            for i in range(g):
                vis = self.vit_encode(clips[b + i][0], rs[b + i])
                self.project_splice(vis, out=buf[i * Lp:(i + 1) * Lp], plan=plans[b + i])
            self.prefill_batch(b, g, buf, L)
            b += g
        done = 1
        while done < max_new_tokens:
            n = min(check_every, max_new_tokens - done) if eos_id is not None else max_new_tokens - done
            self.decode(n)
            done += n
            if eos_id is not None and self.unfinished() == 0:
                break
        return self.outputs()

    # ------------------------------------------------------------------ continuous batching
    def slot_reset(self, slot: int):
        check(self.ctx, self.L.aur_slot_reset(self.ctx, slot, self._stream()), "aur_slot_reset")

    def slot_retire(self, slot: int):
        check(self.ctx, self.L.aur_slot_retire(self.ctx, slot, self._stream()), "aur_slot_retire")

    def slot_collect(self, slot0: int, nslots: int, ids_dev: torch.Tensor, lens_dev: torch.Tensor):
        """Async device copy of the ids / lengths of slots [slot0, slot0 + nslots) (no host synchronisation)."""
        assert ids_dev.dtype == torch.int32 and lens_dev.dtype == torch.int32 and ids_dev.is_contiguous() and lens_dev.is_contiguous()
        assert ids_dev.numel() >= nslots * self._max_new and lens_dev.numel() >= nslots
        check(self.ctx, self.L.aur_slot_collect(self.ctx, slot0, nslots, ids_dev.data_ptr(), lens_dev.data_ptr(), self._stream()),
              "aur_slot_collect")

    def slot_state(self):
        lens = np.zeros(self._batch, np.int32)
        fin = np.zeros(self._batch, np.int32)
        check(self.ctx, self.L.aur_slot_state(self.ctx, lens.ctypes.data_as(C.POINTER(C.c_int32)),
                                              fin.ctypes.data_as(C.POINTER(C.c_int32)), self._stream()), "aur_slot_state")
        return lens, fin

    def caption_stream(self, clips, token_kept_ratio: float, max_new_tokens: int, eos_id: Optional[int] = 2, slots: Optional[int] = None,
                       check_every: int = 16, on_error=None, overlap: Optional[bool] = None, front_cus: int = 16, front_graph: bool = True):
        """Continuous batching over an iterable of (pixel_values, input_ids): up to `slots` (default max_batch) captions are
        in flight; every `check_every` decode steps the finished slots are collected and re-filled with the next clips
        (ViT + projector + prefill into the free slot while the others keep their KV and state).  Yields (index, ids) in
        completion order; each clip's ids equal what it produces alone (batch-invariant kernels).
        on_error(index, exception): a clip whose front end is rejected (shape, context longer than max_ctx, ...) is reported
        and skipped while the captions in flight keep going (the reference harness turns a failing request into an empty
        caption, lmms_eval/models/auroracap.py:511-514); None = raise.
        overlap (default: on when the engine was built with spare_slots > 0; whoever builds the engine decides with
        `streams.overlap_pays` whether a decode of that many slots is K / V-bound enough for it): the front ends of the NEXT clips run ahead on
        their own stream, restricted to `front_cus` CUs of every XCD, into the spare KV sequences while every slot keeps
        decoding (decode is HBM-bound, ViT + prefill MFMA-bound); a freed slot takes over a prepared clip's pages at the next
        check (`prefill_commit`).  front_graph (overlapped mode): each clip's front end is one hipGraph replay, captured per shape
        bucket on first use (`_front_end_replay`); False = ~600 eager launches per clip.  Same ids either way."""
        B = self.max_batch if slots is None else slots
        if not 1 <= B <= self.max_batch:
            raise ValueError(f"slots={B} for an engine built with max_batch={self.max_batch}")
        if overlap is None:
            overlap = self.spare_slots > 0
        if overlap:
            if self.spare_slots < 1:
                raise ValueError("caption_stream(overlap=True) needs an engine built with spare_slots >= 1")
            yield from self._caption_stream_overlapped(clips, token_kept_ratio, max_new_tokens, eos_id, B, check_every, on_error, front_cus,
                                                       front_graph)
            return
        self.begin_batch(B, max_new_tokens, eos_id)
        for s in range(B):
            self.slot_retire(s)
        it = iter(enumerate(clips))
        owner: List[Optional[int]] = [None] * B
        exhausted = False
        while True:
            for s in range(B):                                    # fill every free slot
                while owner[s] is None and not exhausted:
                    nxt = next(it, None)
                    if nxt is None:
                        exhausted = True
                        break
                    idx, (px, ids) = nxt
                    try:
                        r = self.tome_r(token_kept_ratio, px.shape[-2], px.shape[-1])
                        vis = self.vit_encode(px, r)
                        emb, L = self.project_splice(vis, list(ids))
                        self.slot_reset(s)
                        self.prefill(s, emb, L)                   # argument checks come before any enqueue: a rejection leaves the slot idle
                        owner[s] = idx
                    except (ValueError, IndexError, AssertionError, _lib.AuroraHipError) as e:
                        if on_error is None:
                            raise
                        on_error(idx, e)
            if all(o is None for o in owner):
                return
            self.decode(check_every)
            lens, fin = self.slot_state()
            done = [s for s in range(B) if owner[s] is not None and fin[s]]
            if done:
                out = self.outputs()
                for s in done:
                    yield owner[s], out[s]
                    owner[s] = None

    def _masked_streams(self, front_cus: int):
        """(front-end stream on `front_cus` CUs of every XCD, decode stream on the other CUs); shared per process (streams.py)."""
        from .streams import shared_cu_masked_stream
        if not hasattr(self, "_masked"):
            self._masked = {}
        if front_cus not in self._masked:
            self._masked[front_cus] = (shared_cu_masked_stream(front_cus, device=self.dev),
                                       shared_cu_masked_stream(32 - front_cus, from_top=True, device=self.dev))
        return self._masked[front_cus]

    def _front_end_replay(self, px: torch.Tensor, ids: List[int], r: int, seq: int):
        """ViT + ToMe + projector / splice + staged prefill of ONE clip into KV sequence `seq` as a single hipGraph replay on torch's
        current stream.  Graphs are captured per shape bucket - (frames, H, W, r, prompt structure, seq) - on first use and kept (at most
        `front_graph_cap`, least recently used evicted): the per-bucket capture of the reference tree's serving engine
        (src/sglang/python/sglang/srt/model_executor/cuda_graph_runner.py:163-279).  Returns (embeds, seq_len) for `prefill_commit`;
        `embeds` is the bucket's static buffer: the caller orders the next replay of the same bucket (same `seq`) after that commit."""
        if px.dim() != 4 or px.shape[1] != self.v.get("num_channels", 3):
            raise ValueError(f"pixel_values must be [frames, {self.v.get('num_channels', 3)}, H, W], got {tuple(px.shape)}")
        F, H, W = int(px.shape[0]), int(px.shape[-2]), int(px.shape[-1])
        t0 = (H // self.v["patch_size"]) * (W // self.v["patch_size"]) + 1
        if F < 1 or F > self.c.max_frames or t0 < 2 or t0 > (self.max_image // self.v["patch_size"]) ** 2 + 1:
            raise ValueError(f"clip of {F} frames of {H}x{W} outside this engine (max_frames {self.c.max_frames}, max_image {self.max_image})")
        n_kept = tokens_at_layer(t0, r, self.v["num_hidden_layers"] - 1) - 1
        plan = self.splice_plan(ids, F, n_kept)
        if plan["seq_len"] < 1 or plan["seq_len"] + self._max_new > self.c.max_ctx:
            raise ValueError(f"prompt of {plan['seq_len']} positions + {self._max_new} new tokens exceeds max_ctx {self.c.max_ctx}")
        key = (F, H, W, r, seq, plan["seq_len"], plan["used"], plan["nvis"], plan["ntext"])
        cache = self._front_graphs
        fg = cache.pop(key, None)
        if fg is None:
            while len(cache) >= self.front_graph_cap:             # evict the least recently used bucket once its last replay has run
                _, old = next(iter(cache.items()))
                cache.pop(next(iter(cache)))
                if old._last is not None:
                    old._last.synchronize()
                old.close()
            fg = FrontEndGraph(self, 1, F, H, W, r, plan, seq0=seq)
            fg._last = None
        cache[key] = fg                                           # most recently used last
        fg.load(self._h(px), [plan])
        fg.launch()
        fg._last = torch.cuda.Event()
        fg._last.record(torch.cuda.current_stream(self.dev))
        return fg.embeds, plan["seq_len"]

    def _caption_stream_overlapped(self, clips, token_kept_ratio, max_new_tokens, eos_id, B, check_every, on_error, front_cus, front_graph=True):
        from collections import deque
        sD = torch.cuda.current_stream(self.dev)
        sF, sDm = self._masked_streams(front_cus)
        self.begin_batch(B, max_new_tokens, eos_id)
        for s in range(B):
            self.slot_retire(s)
        it = iter(enumerate(clips))
        owner: List[Optional[int]] = [None] * B
        free_seqs = list(range(self.max_batch, self.max_batch + self.spare_slots))     # spare KV sequences holding no prepared clip
        reusable = {}                                             # sequence -> event after which its (exchanged) pages may be rewritten
        staged = deque()                                          # prepared clips: (index, sequence, embeds, L, event)
        exhausted = False
        sF.wait_stream(sD)
        self.set_option("gemm_max_wgs", 8 * front_cus)            # the persistent GEMM sizes its grid to the front end's CUs
        k_masked = max(1, int(round(0.8 * check_every)))
        try:
            while True:
                # 1. slots that are free take over prepared clips (decode stream, between two decode calls)
                for s in range(B):
                    if owner[s] is None and staged:
                        idx, seq, emb, L, ev = staged.popleft()
                        sD.wait_event(ev)
                        emb.record_stream(sD)
                        self.prefill_commit(s, 1, seq, emb, L)
                        evc = torch.cuda.Event()
                        evc.record(sD)
                        reusable[seq] = evc
                        free_seqs.append(seq)
                        owner[s] = idx
                # 2. one decode chunk, enqueued BEFORE the next front ends (a few hundred launches on the host: the decode must not
                #    sit behind them); while a front end will be in flight its first part runs on the complementary CU mask
                decoding = any(o is not None for o in owner)
                if decoding:
                    k1 = min(check_every, k_masked) if (free_seqs and not exhausted) else 0
                    if k1 > 0:
                        sDm.wait_stream(sD)
                        with torch.cuda.stream(sDm):
                            # on half of the CUs the QKV / gate-up projections launch half as many workgroups with twice the tiles each
                            # (bitwise the same tokens; engines of <= 64 slots ignore it)
                            self.set_option("decode_half_grid", 1)
                            try:
                                self.decode(k1)
                            finally:                                # a failed decode must not leave the ctx replaying the half-grid graph
                                self.set_option("decode_half_grid", 0)
                        sD.wait_stream(sDm)
                    if check_every - k1 > 0:
                        self.decode(check_every - k1)
                # 3. prepare the next clips into every free spare sequence (front-end stream: runs beside the decode above)
                while free_seqs and not exhausted:
                    # the iterator may be lazy (the lmms-eval adaptor decodes and preprocesses a clip when asked): whatever it
                    # enqueues on torch's current stream must be ordered before the front end that reads it -> pull the clip
                    # with the front-end stream current, so its kernels and allocations belong to that stream
                    with torch.cuda.stream(sF):
                        nxt = next(it, None)
                    if nxt is None:
                        exhausted = True
                        break
                    idx, (px, ids) = nxt
                    seq = free_seqs.pop()
                    try:
                        with torch.cuda.stream(sF):
                            if torch.is_tensor(px) and px.is_cuda:
                                px.record_stream(sF)              # allocated on another stream (a prebuilt clip): not reusable before sF is done with it
                            if seq in reusable:
                                sF.wait_event(reusable.pop(seq))
                            r = self.tome_r(token_kept_ratio, px.shape[-2], px.shape[-1])
                            if front_graph:
                                emb, L = self._front_end_replay(px, list(ids), r, seq)      # one hipGraph replay per clip (captured per shape bucket)
                            else:
                                vis = self.vit_encode(px, r)
                                emb, L = self.project_splice(vis, list(ids))
                                self.prefill_stage(seq, 1, emb, L)    # argument checks come before any enqueue
                            ev = torch.cuda.Event()
                            ev.record(sF)
                        staged.append((idx, seq, emb, L, ev))
                    except (ValueError, IndexError, AssertionError, _lib.AuroraHipError) as e:
                        free_seqs.append(seq)
                        if on_error is None:
                            raise
                        on_error(idx, e)
                if not decoding:
                    if not staged:
                        return
                    continue                                      # nothing decoding yet: commit what was just prepared
                lens, fin = self.slot_state()
                done = [s for s in range(B) if owner[s] is not None and fin[s]]
                if done:
                    out = self.outputs()
                    for s in done:
                        yield owner[s], out[s]
                        owner[s] = None
        finally:
            sD.wait_stream(sF)
            self.set_option("gemm_max_wgs", 0)

    # ------------------------------------------------------------------ captured call sequences (hipGraph)
    def graph_capture(self, fn):
        """Record the engine calls `fn()` enqueues on torch's current stream into a hipGraph (aur_graph_begin / aur_graph_end) and
        return (handle, number of nodes).  `fn` must only make enqueue-only engine calls on pre-allocated tensors: a torch
        allocation, a synchronisation or `decode` inside it breaks the capture (raised here)."""
        st = self._stream()
        check(self.ctx, self.L.aur_graph_begin(self.ctx, st), "aur_graph_begin")
        gid, nodes = C.c_int32(0), C.c_int64(0)
        try:
            fn()
        except BaseException:
            self.L.aur_graph_end(self.ctx, st, C.byref(gid), C.byref(nodes))      # always close the capture; its status is secondary
            if gid.value:
                self.L.aur_graph_destroy(self.ctx, gid.value)
            raise
        check(self.ctx, self.L.aur_graph_end(self.ctx, st, C.byref(gid), C.byref(nodes)), "aur_graph_end")
        return gid.value, nodes.value

    def graph_launch(self, gid: int):
        check(self.ctx, self.L.aur_graph_launch(self.ctx, gid, self._stream()), "aur_graph_launch")

    def graph_destroy(self, gid: int):
        check(self.ctx, self.L.aur_graph_destroy(self.ctx, gid), "aur_graph_destroy")

    def decode_stamps(self, cap: int = 4096):
        """(microseconds per step of the stamped attention launch, oldest first; total steps stamped) - option
        "decode_stamp_layer"; synchronises."""
        buf = (C.c_double * cap)()
        n, tot = C.c_int32(0), C.c_int64(0)
        check(self.ctx, self.L.aur_decode_stamps_read(self.ctx, buf, cap, C.byref(tot), C.byref(n), self._stream()), "aur_decode_stamps_read")
        return np.asarray(buf[:n.value], dtype=np.float64), tot.value

    # ------------------------------------------------------------------ kernel-level entry points (tests)
    def tome_step(self, metric: torch.Tensor, x: torch.Tensor, size: Optional[torch.Tensor], r: int):
        F, t, c = metric.shape
        d = x.shape[-1]
        rl = max(0, min(r, (t - 1) // 2))
        ta = (t + 1) // 2
        m = self._f(metric)
        xh = self._h(x)
        s = self._f(size.reshape(F, t)) if size is not None else None
        xo = torch.empty(F, t - rl, d, dtype=torch.float16, device=self.dev)
        so = torch.empty(F, t - rl, dtype=torch.float32, device=self.dev)
        ni = torch.zeros(F, ta, dtype=torch.int32, device=self.dev)
        un = torch.zeros(F, max(ta - rl, 1), dtype=torch.int32, device=self.dev)
        sr = torch.zeros(F, max(rl, 1), dtype=torch.int32, device=self.dev)
        ds = torch.zeros(F, max(rl, 1), dtype=torch.int32, device=self.dev)
        check(self.ctx, self.L.aur_tome_step(self.ctx, m.data_ptr(), xh.data_ptr(), s.data_ptr() if s is not None else None,
                                             F, t, c, d, r, xo.data_ptr(), so.data_ptr(), ni.data_ptr(), un.data_ptr(),
                                             sr.data_ptr(), ds.data_ptr(), self._stream()), "aur_tome_step")
        torch.cuda.synchronize()
        return xo, so, dict(r=rl, node_idx=ni, unm_idx=un[:, : ta - rl], src_idx=sr[:, :rl], dst_idx=ds[:, :rl])

    def linear(self, a: torch.Tensor, w: torch.Tensor, bias: Optional[torch.Tensor] = None, act: int = 0,
               resid: Optional[torch.Tensor] = None) -> torch.Tensor:
        n, k = w.shape
        npad = _rup(n, 256)
        wp = self.pack(w, npad, k)
        b = self._bias(bias, npad) if bias is not None else None
        ah = self._h(a)
        rh = self._h(resid) if resid is not None else None
        out = torch.empty(ah.shape[0], n, dtype=torch.float16, device=self.dev)
        check(self.ctx, self.L.aur_linear(self.ctx, ah.data_ptr(), ah.shape[0], k, wp.data_ptr(), npad, n,
                                          b.data_ptr() if b is not None else None, act, rh.data_ptr() if rh is not None else None,
                                          out.data_ptr(), self._stream()), "aur_linear")
        torch.cuda.synchronize()
        return out

    def linear_skinny(self, a: torch.Tensor, w: torch.Tensor) -> torch.Tensor:
        n, k = w.shape
        npad = _rup(n, 256)
        wp = self.pack(w, npad, k)
        ah = self._h(a)
        out = torch.empty(ah.shape[0], n, dtype=torch.float32, device=self.dev)
        check(self.ctx, self.L.aur_linear_skinny(self.ctx, ah.data_ptr(), ah.shape[0], k, wp.data_ptr(), npad, n, out.data_ptr(),
                                                 self._stream()), "aur_linear_skinny")
        torch.cuda.synchronize()
        return out

    def layernorm(self, x, w, b, eps):
        xh = self._h(x)
        y = torch.empty_like(xh)
        wf, bf = self._f(w), self._f(b)
        check(self.ctx, self.L.aur_layernorm(self.ctx, xh.data_ptr(), xh.shape[0], xh.shape[1], wf.data_ptr(), bf.data_ptr(),
                                             eps, y.data_ptr(), self._stream()), "aur_layernorm")
        torch.cuda.synchronize()
        return y

    def rmsnorm(self, x, w, eps):
        xh = self._h(x)
        y = torch.empty_like(xh)
        wf = self._f(w)
        check(self.ctx, self.L.aur_rmsnorm(self.ctx, xh.data_ptr(), xh.shape[0], xh.shape[1], wf.data_ptr(), eps, y.data_ptr(),
                                           self._stream()), "aur_rmsnorm")
        torch.cuda.synchronize()
        return y

    def vit_layer(self, layer: int, x: torch.Tensor, size: Optional[torch.Tensor], r: int):
        F, t, D = x.shape
        hd = D // self.v["num_attention_heads"]
        rl = max(0, min(r, (t - 1) // 2))
        ta = (t + 1) // 2
        xh = self._h(x)
        s = self._f(size.reshape(F, t)) if size is not None else None
        xo = torch.empty(F, t - rl, D, dtype=torch.float16, device=self.dev)
        so = torch.empty(F, t - rl, dtype=torch.float32, device=self.dev)
        me = torch.zeros(F, t, hd, dtype=torch.float32, device=self.dev)
        ni = torch.zeros(F, ta, dtype=torch.int32, device=self.dev)
        un = torch.zeros(F, max(ta - rl, 1), dtype=torch.int32, device=self.dev)
        sr = torch.zeros(F, max(rl, 1), dtype=torch.int32, device=self.dev)
        ds = torch.zeros(F, max(rl, 1), dtype=torch.int32, device=self.dev)
        check(self.ctx, self.L.aur_vit_layer(self.ctx, layer, xh.data_ptr(), s.data_ptr() if s is not None else None, F, t, r,
                                             xo.data_ptr(), so.data_ptr(), me.data_ptr(), ni.data_ptr(), un.data_ptr(),
                                             sr.data_ptr(), ds.data_ptr(), self._stream()), "aur_vit_layer")
        torch.cuda.synchronize()
        return xo, so, me, dict(r=rl, node_idx=ni, unm_idx=un[:, : ta - rl], src_idx=sr[:, :rl], dst_idx=ds[:, :rl])

    def set_option(self, name: str, value: int):
        check(self.ctx, self.L.aur_set_option(self.ctx, name.encode(), int(value)), "aur_set_option")

    def microbench(self, kernel: str, iters: int = 200) -> float:
        us = C.c_double(0)
        check(self.ctx, self.L.aur_microbench(self.ctx, kernel.encode(), iters, C.byref(us), self._stream()), "aur_microbench")
        return us.value

    # ------------------------------------------------------------------ profiling
    def profile(self, on: bool):
        check(self.ctx, self.L.aur_profile_enable(self.ctx, int(on)), "aur_profile_enable")

    def profile_read(self, stage: str):
        ms, n = C.c_double(0), C.c_int64(0)
        check(self.ctx, self.L.aur_profile_read(self.ctx, stage.encode(), C.byref(ms), C.byref(n)), "aur_profile_read")
        return ms.value, n.value


class FrontEndGraph:
    """The front end of a GROUP of `group` equal-shape clips - ViT + per-layer ToMe, projector + splice, and the staged prefill
    of the group into KV sequences [seq0, seq0 + group) - captured once into a hipGraph and replayed with one call.

    The reference tree's own serving engine captures one graph per shape bucket and copies each request's inputs into the graph's
    static buffers before the replay (src/sglang/python/sglang/srt/model_executor/cuda_graph_runner.py:163-279); the bucket here is
    (group, frames, H, W, r, prompt structure, seq0), the static buffers are `pixels`, the three splice-plan arrays and `embeds`.
    A front end is ~600 launches (31 ViT layers x 10 + 32 prefill layers x 8 + splices): replayed from the graph the host submits it
    in one call, so a serving loop can submit it just before the device can start it (bounded run-ahead: host-observed TTFT stays
    close to the device's) without starving the front-end stream.

        fg = FrontEndGraph(eng, group, frames, H, W, r, plan, seq0, embeds=emb)     # on the stream that will replay it
        fg.load(pixels, plans); fg.launch()                                        # per group; then eng.prefill_commit(.., fg.embeds, ..)
    """

    def __init__(self, eng: "AuroraCapEngine", group: int, frames: int, height: int, width: int, r: int, plan: dict, seq0: int,
                 embeds: Optional[torch.Tensor] = None, prefill: str = "stage", slot0: int = 0):
        v, d = eng.v, eng.l["hidden_size"]
        self.eng, self.group, self.frames, self.r, self.seq0 = eng, group, frames, r, seq0
        self.seq_len = plan["seq_len"]
        self.mseq = _rup(self.seq_len, 32)
        t0 = (height // v["patch_size"]) * (width // v["patch_size"]) + 1
        self.n_kept = tokens_at_layer(t0, r, v["num_hidden_layers"] - 1) - 1
        assert not np.isscalar(plan["n_kept"]) or plan["n_kept"] == self.n_kept
        dev = eng.dev
        self.pixels = torch.zeros(group * frames, v.get("num_channels", 3), height, width, dtype=torch.float16, device=dev)
        self.vis = torch.empty(group * frames, self.n_kept, v["hidden_size"], dtype=torch.float16, device=dev)
        self.embeds = embeds if embeds is not None else torch.zeros(group * self.mseq, d, dtype=torch.float16, device=dev)
        assert self.embeds.shape[0] >= group * self.mseq and self.embeds.is_contiguous()
        # the bucket's splice plans: same structure as `plan` (row maps of one prompt), contents replaced per clip by load()
        self.used, self.nvis, self.ntext = plan["used"], plan["nvis"], plan["ntext"]
        self.vis_rows = torch.zeros(group, max(self.nvis, 1), dtype=torch.int32, device=dev)
        self.text_ids = torch.zeros(group, max(self.ntext, 1), dtype=torch.int32, device=dev)
        self.text_rows = torch.zeros(group, max(self.ntext, 1), dtype=torch.int32, device=dev)
        self._plans = [dict(seq_len=self.seq_len, used=self.used, n_kept=plan["n_kept"], nvis=self.nvis, ntext=self.ntext,
                            vis_rows=self.vis_rows[j], text_ids=self.text_ids[j], text_rows=self.text_rows[j]) for j in range(group)]
        eng.interpolated_pos(height, width)                       # the position table of this input size exists before the capture
        for j in range(group):
            self.vis_rows[j].copy_(plan["vis_rows"])
            self.text_ids[j].copy_(plan["text_ids"])
            self.text_rows[j].copy_(plan["text_rows"])

        def run():
            eng.vit_encode(self.pixels, r, out=self.vis)
            for j in range(group):
                eng.project_splice(self.vis[j * frames:(j + 1) * frames], plan=self._plans[j], out=self.embeds[j * self.mseq:(j + 1) * self.mseq])
            if prefill == "stage":
                eng.prefill_stage(seq0, group, self.embeds, self.seq_len)
            elif prefill == "batch":
                eng.prefill_batch(slot0, group, self.embeds, self.seq_len)
            else:
                assert prefill == "none"

        run()                                                      # once eagerly: lazily set kernel attributes must not happen inside a capture
        self.gid, self.nodes = eng.graph_capture(run)

    def load(self, pixels: torch.Tensor, plans: Optional[Sequence[dict]] = None, *, vis_rows=None, text_ids=None, text_rows=None):
        """Copy a group's inputs into the graph's static buffers (device-to-device on torch's current stream).  plans: the clips'
        `splice_plan`s (same structure as the bucket's), or stacked [group, n] int32 arrays through the keyword arguments."""
        self.pixels.copy_(pixels.reshape(self.pixels.shape), non_blocking=True)
        if plans is not None:
            assert len(plans) == self.group
            for j, pl in enumerate(plans):
                if (pl["seq_len"], pl["used"], pl["nvis"], pl["ntext"]) != (self.seq_len, self.used, self.nvis, self.ntext):
                    raise ValueError("splice plan of another shape bucket")
                self.vis_rows[j, :self.nvis].copy_(pl["vis_rows"], non_blocking=True)
                self.text_ids[j, :self.ntext].copy_(pl["text_ids"], non_blocking=True)
                self.text_rows[j, :self.ntext].copy_(pl["text_rows"], non_blocking=True)
        if vis_rows is not None:
            self.vis_rows.copy_(vis_rows, non_blocking=True)
        if text_ids is not None:
            self.text_ids.copy_(text_ids, non_blocking=True)
        if text_rows is not None:
            self.text_rows.copy_(text_rows, non_blocking=True)

    def launch(self):
        self.eng.graph_launch(self.gid)

    def close(self):
        if self.gid:
            self.eng.graph_destroy(self.gid)
            self.gid = 0
