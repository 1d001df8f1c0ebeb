"""Run under the host-sanitizer build of libaurora_hip (tests/test_asan_host.py sets LD_PRELOAD / AURORA_HIP_SO / ASAN_OPTIONS).
Without arguments: every entry point that needs no GPU.  With `gpu`: a tiny engine through the whole path, both schedules."""
import ctypes as C
import sys

import numpy as np

from aurora_amd import _lib


def host_only():
    L = _lib.lib()
    assert b"aurora_hip" in L.aur_version()
    for ratio in (1.0, 0.8, 0.3, 0.2, 0.01):
        for layers in (4, 24, 32):
            r = L.aur_tome_r(378, 378, 14, ratio, layers)
            t = L.aur_tokens_at_layer(730, r, layers - 1)
            assert 1 <= t <= 730
    for h, w in ((378, 378), (720, 1280), (1080, 1920), (90, 70), (2160, 3840)):
        n = L.aur_preprocess_plan_len(h, w, 378)
        assert n > 0
        plan = np.empty(n, np.int32)
        assert L.aur_preprocess_plan(h, w, 378, plan.ctypes.data_as(C.c_void_p), n) == 0
        assert L.aur_preprocess_plan(h, w, 378, plan.ctypes.data_as(C.c_void_p), n - 1) != 0       # short buffer: rejected, not overrun
    ctx = C.c_void_p()
    bad = _lib.AurConfig(vit_hidden=100, vit_heads=3)                                               # invalid dims: argument error path
    assert L.aur_create(C.byref(bad), C.byref(ctx)) == -1 and len(L.aur_last_error(None)) > 0
    assert L.aur_create(None, C.byref(ctx)) == -1
    print("host-only ABI walk ok")


def with_gpu():
    import torch
    from aurora_amd.engine import AuroraCapEngine
    from tests.util import rand_llm_weights, rand_proj_weights, rand_vit_weights
    vcfg = dict(hidden_size=64, num_attention_heads=4, num_hidden_layers=4, intermediate_size=128, patch_size=14, image_size=56,
                hidden_act="quick_gelu", layer_norm_eps=1e-5)
    lcfg = dict(hidden_size=128, num_attention_heads=4, num_hidden_layers=2, intermediate_size=256, vocab_size=320, rms_norm_eps=1e-5,
                rope_theta=1e4, rope_factor=4.0)
    w = {"vit": rand_vit_weights(vcfg, 1), "projector": rand_proj_weights(64, 128, 2), "llm": rand_llm_weights(lcfg, 3)}
    eng = AuroraCapEngine({"vit": vcfg, "llm": lcfg}, w, max_frames=4, max_batch=3, max_ctx=256, max_new_tokens=12, spare_slots=2)
    try:
        g = torch.Generator().manual_seed(3)
        clips = [(torch.randn(1 + i % 3, 3, 56, 56, generator=g).half(), [1, 9] + [-200, 30] * (1 + i % 3) + [40]) for i in range(7)]
        alone = [eng.caption_ids(px, ids, 0.5, 12, eos_id=None) for px, ids in clips]
        assert eng.caption_batch(clips[:3], 0.5, 12, eos_id=None) == alone[:3]
        for overlap in (False, True):
            got = dict(eng.caption_stream(clips, 0.5, 12, eos_id=None, check_every=4, overlap=overlap))
            assert [got[i] for i in range(7)] == alone, overlap
        try:
            eng.prefill(5, torch.zeros(32, 128, dtype=torch.float16, device="cuda"), 10)           # slot outside the batch: error path
            raise AssertionError("expected an argument error")
        except _lib.AuroraHipError:
            pass
    finally:
        eng.close()
    print("engine walk under the host sanitizers ok")


if __name__ == "__main__":
    host_only()
    if len(sys.argv) > 1 and sys.argv[1] == "gpu":
        with_gpu()
