"""ctypes binding of libaurora_hip.so (the C ABI in include/aurora_hip.h).

The product path has NO CPU fallback: if the HIP library is missing or fails to load, importing this
module's `lib()` raises.  Nothing here imports `oracle/`.
"""
from __future__ import annotations

import ctypes as C
import os

HERE = os.path.dirname(os.path.abspath(__file__))
# AURORA_HIP_SO selects another build of the same library (the host-sanitizer build of aurora_amd/build.py, tests/test_asan_host.py)
SO_PATH = os.environ.get("AURORA_HIP_SO") or os.path.join(HERE, "libaurora_hip.so")

AUR_ACT_QUICK_GELU = 1
AUR_ACT_GELU = 2
AUR_ACT_SILU = 4
AUR_ACT_RELU = 5
AUR_ACT_GELU_TANH = 6
# transformers' ACT2FN names (modeling_projector.py:27 `ACT2FN[config.hidden_act]`) the kernels' epilogues implement; anything else raises
ACT_BY_NAME = {"quick_gelu": AUR_ACT_QUICK_GELU, "gelu": AUR_ACT_GELU, "silu": AUR_ACT_SILU, "swish": AUR_ACT_SILU, "relu": AUR_ACT_RELU,
               "gelu_new": AUR_ACT_GELU_TANH, "gelu_pytorch_tanh": AUR_ACT_GELU_TANH}


class AurConfig(C.Structure):
    _fields_ = [
        ("vit_hidden", C.c_int32), ("vit_heads", C.c_int32), ("vit_layers", C.c_int32), ("vit_mlp", C.c_int32),
        ("vit_patch", C.c_int32), ("vit_image", C.c_int32), ("vit_channels", C.c_int32), ("vit_act", C.c_int32),
        ("vit_ln_eps", C.c_float),
        ("llm_hidden", C.c_int32), ("llm_heads", C.c_int32), ("llm_layers", C.c_int32), ("llm_mlp", C.c_int32),
        ("llm_vocab", C.c_int32),
        ("llm_rms_eps", C.c_float), ("rope_theta", C.c_float), ("rope_factor", C.c_float),
        ("max_frames", C.c_int32), ("max_batch", C.c_int32), ("max_ctx", C.c_int32), ("max_new_tokens", C.c_int32),
        ("page_tokens", C.c_int32), ("use_graph", C.c_int32), ("num_banks", C.c_int32), ("vit_native_image", C.c_int32),
        ("spare_slots", C.c_int32), ("proj_depth", C.c_int32), ("proj_act", C.c_int32),
    ]


class AuroraHipError(RuntimeError):
    pass


_P = C.c_void_p
_I = C.c_int32
_L = C.c_int64
_IP = C.POINTER(C.c_int32)

# name -> (restype, argtypes); MUST list every symbol declared in include/aurora_hip.h
SIGNATURES = {
    "aur_create": (C.c_int, [C.POINTER(AurConfig), C.POINTER(_P)]),
    "aur_destroy": (None, [_P]),
    "aur_last_error": (C.c_char_p, [_P]),
    "aur_version": (C.c_char_p, []),
    "aur_set_tensor": (C.c_int, [_P, C.c_char_p, _P, _L]),
    "aur_workspace_bytes": (_L, [_P]),
    "aur_kv_pool_bytes": (_L, [_P]),
    "aur_set_workspace": (C.c_int, [_P, _P, _L]),
    "aur_set_kv_pool": (C.c_int, [_P, _P, _L]),
    "aur_finalize": (C.c_int, [_P, _P]),
    "aur_pack_linear": (C.c_int, [_P, _P, _I, _I, _I, _P, _I, _I, _P, _P]),
    "aur_tome_r": (_I, [_I, _I, _I, C.c_double, _I]),
    "aur_tokens_at_layer": (_I, [_I, _I, _I]),
    "aur_vit_encode": (C.c_int, [_P, _P, _I, _I, _P, _IP, _P]),
    "aur_vit_pos_interp": (C.c_int, [_P, _I, _I, _P, _P]),
    "aur_vit_encode_hw": (C.c_int, [_P, _P, _I, _I, _I, _P, _I, _P, _IP, _P]),
    "aur_project_splice": (C.c_int, [_P, _P, _I, _P, _P, _P, _I, _I, _P, _P]),
    "aur_begin_batch": (C.c_int, [_P, _I, _I, _I, _P]),
    "aur_select_bank": (C.c_int, [_P, _I]),
    "aur_llm_prefill": (C.c_int, [_P, _I, _P, _I, _P]),
    "aur_llm_prefill_batch": (C.c_int, [_P, _I, _I, _P, _I, _P]),
    "aur_llm_prefill_stage": (C.c_int, [_P, _I, _I, _P, _I, _P]),
    "aur_llm_prefill_commit": (C.c_int, [_P, _I, _I, _I, _P, _I, _P]),
    "aur_llm_decode": (C.c_int, [_P, _I, _P]),
    "aur_get_outputs": (C.c_int, [_P, _IP, _IP, _P]),
    "aur_unfinished": (C.c_int, [_P, _IP, _P]),
    "aur_slot_reset": (C.c_int, [_P, _I, _P]),
    "aur_slot_retire": (C.c_int, [_P, _I, _P]),
    "aur_slot_state": (C.c_int, [_P, _IP, _IP, _P]),
    "aur_slot_collect": (C.c_int, [_P, _I, _I, _P, _P, _P]),
    "aur_graph_begin": (C.c_int, [_P, _P]),
    "aur_graph_end": (C.c_int, [_P, _P, _IP, C.POINTER(_L)]),
    "aur_graph_launch": (C.c_int, [_P, _I, _P]),
    "aur_graph_destroy": (C.c_int, [_P, _I]),
    "aur_decode_stamps_read": (C.c_int, [_P, C.POINTER(C.c_double), _I, C.POINTER(_L), _IP, _P]),
    "aur_tome_step": (C.c_int, [_P, _P, _P, _P, _I, _I, _I, _I, _I, _P, _P, _P, _P, _P, _P, _P]),
    "aur_linear": (C.c_int, [_P, _P, _I, _I, _P, _I, _I, _P, _I, _P, _P, _P]),
    "aur_linear_skinny": (C.c_int, [_P, _P, _I, _I, _P, _I, _I, _P, _P]),
    "aur_layernorm": (C.c_int, [_P, _P, _I, _I, _P, _P, C.c_float, _P, _P]),
    "aur_rmsnorm": (C.c_int, [_P, _P, _I, _I, _P, C.c_float, _P, _P]),
    "aur_vit_layer": (C.c_int, [_P, _I, _P, _P, _I, _I, _I, _P, _P, _P, _P, _P, _P, _P, _P]),
    "aur_copy_logits": (C.c_int, [_P, _P, _P]),
    "aur_set_option": (C.c_int, [_P, C.c_char_p, _L]),
    "aur_microbench": (C.c_int, [_P, C.c_char_p, _I, C.POINTER(C.c_double), _P]),
    "aur_profile_enable": (C.c_int, [_P, _I]),
    "aur_profile_read": (C.c_int, [_P, C.c_char_p, C.POINTER(C.c_double), C.POINTER(_L)]),
    "aur_preprocess_plan_len": (_L, [_I, _I, _I]),
    "aur_preprocess_plan": (C.c_int, [_I, _I, _I, _P, _L]),
    "aur_preprocess_tmp_bytes": (_L, [_I, _I, _I, _I]),
    "aur_preprocess_frames": (C.c_int, [_P, _I, _I, _I, _I, _P, _P, _P, _P, _P]),
}

_lib = None


def lib():
    """Load the HIP library (fails loudly - there is no fallback path)."""
    global _lib
    if _lib is None:
        if not os.path.exists(SO_PATH):
            raise AuroraHipError(
                f"{SO_PATH} is missing: build it with `python -m aurora_amd.build` (hipcc --offload-arch=gfx950). "
                "aurora_amd has no CPU fallback.")
        # One HIP runtime per process: torch ships its own libamdhip64 and owns the device memory and streams this
        # library is handed, so it must be resident before the .so resolves its libamdhip64 dependency. Loaded the other
        # way round the .so binds /opt/rocm's copy, torch then brings a second runtime, and the first kernel-attribute
        # call in aur_create reports "no ROCm-capable device".
        import torch  # noqa: F401
        l = C.CDLL(SO_PATH)
        for name, (res, args) in SIGNATURES.items():
            fn = getattr(l, name)          # AttributeError if the .so does not export it
            fn.restype = res
            fn.argtypes = args
        _lib = l
    return _lib


def check(ctx, rc: int, what: str = ""):
    if rc != 0:
        msg = lib().aur_last_error(ctx)
        raise AuroraHipError(f"{what} failed (status {rc}): {msg.decode() if msg else ''}")
