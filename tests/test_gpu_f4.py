"""GPU: non-native input sizes (interpolated position table, rectangular patch grid) and slow-fast mode through the
C ABI, against the oracle."""
import numpy as np
import pytest
import torch

from oracle import aurora_oracle as O
from tests.test_gpu_llm import LLM_CFGS
from tests.util import golden, rand_llm_weights, rand_proj_weights, rand_vit_weights, rel_l2

pytestmark = pytest.mark.gpu

VCFG = dict(hidden_size=64, num_attention_heads=4, num_hidden_layers=4, intermediate_size=128, patch_size=14,
            image_size=84, hidden_act="quick_gelu", layer_norm_eps=1e-5)


def vit_engine(frames=3):
    from aurora_amd.engine import AuroraCapEngine
    w = rand_vit_weights(VCFG, 4)
    return AuroraCapEngine({"vit": VCFG, "llm": None}, {"vit": w}, max_frames=frames, max_batch=1, max_ctx=128, max_new_tokens=8), w


def test_interpolated_position_table_matches_reference_fixture():
    """The engine's table for a non-native grid, computed from the fp16-stored checkpoint table, against the table the
    reference's interpolate_pos_encoding produced from the same (fp16-representable) values."""
    from aurora_amd.engine import AuroraCapEngine
    g11 = golden("g11_pos_interp.npz")
    pos = torch.from_numpy(g11["tiny.pos"]).half().float()
    cfg = dict(VCFG, image_size=56)
    w = rand_vit_weights(cfg, 4)
    w["position_embedding.weight"] = pos
    eng = AuroraCapEngine({"vit": cfg, "llm": None}, {"vit": w}, max_frames=1, max_batch=1, max_ctx=128, max_new_tokens=8)
    try:
        assert eng.interpolated_pos(56, 56) is None
        worst = 0
        for h, wd in [(56, 84), (42, 56), (70, 70), (28, 98), (84, 56), (98, 28), (14, 14), (112, 42)]:
            want = O.interpolate_pos_encoding(pos, h, wd, 14)
            got = eng.interpolated_pos(h, wd).float().cpu()
            assert got.shape == want.shape
            # pos_interp_kernel (vit.hip) restates ATen's bicubic expression by expression: against torch's own fp32 result rounded to
            # fp16, at most one fp16 step apart where the two fp32 values straddle a rounding boundary, equal almost everywhere
            w16 = want.half().float()
            step = torch.maximum(w16.abs(), torch.tensor(2.0 ** -14)) * 2.0 ** -10
            assert ((got - w16).abs() <= step).all()
            assert (got[0] == pos[0]).all()                                   # the class row is kept (aurora.py:947)
            worst = max(worst, float((got != w16).float().mean()))
        assert worst < 0.01, worst
            # and the oracle itself is pinned to the reference on the unrounded table (tests/test_f4_oracle.py)
    finally:
        eng.close()


@pytest.mark.parametrize("h,w", [(56, 84), (84, 56), (42, 56), (70, 70), (28, 98), (84, 84), (60, 75)])
def test_vit_encode_non_native_sizes(h, w):
    eng, wts = vit_engine()
    try:
        px = torch.randn(3, 3, h, w, generator=torch.Generator().manual_seed(h * 100 + w)).half().float()
        out = eng.vit_encode(px, eng.tome_r(1.0, h, w)).float().cpu()
        ref = O.vit_features(px, wts, VCFG, 1.0)
        assert out.shape == ref.shape == (3, (h // 14) * (w // 14), 64)
        assert rel_l2(out, ref) < 5e-3
        r = eng.tome_r(0.5, h, w)
        assert r == O.tome_r(h, w, 14, 0.5, 4)
        out = eng.vit_encode(px, r).float().cpu()
        ref = O.vit_features(px, wts, VCFG, 0.5, q=O.fp16_storage)
        assert out.shape == ref.shape
        assert rel_l2(out.mean(1), ref.mean(1)) < 1e-2
    finally:
        eng.close()


def test_vit_encode_rejects_more_tokens_than_the_engine_holds():
    eng, _ = vit_engine()
    try:
        with pytest.raises(ValueError):
            eng.vit_encode(torch.zeros(1, 3, 98, 98), 0)
        with pytest.raises(ValueError):
            eng.vit_encode(torch.zeros(1, 3, 10, 84), 0)                       # less than one patch row
    finally:
        eng.close()


def test_slowfast_forward_against_oracle():
    from aurora_amd.engine import AuroraCapEngine
    from aurora_amd.model import AuroraModel
    vcfg = dict(VCFG, image_size=56)
    lcfg = LLM_CFGS["hd32"]
    w = {"vit": rand_vit_weights(vcfg, 1), "projector": rand_proj_weights(64, lcfg["hidden_size"], 2), "llm": rand_llm_weights(lcfg, 3)}
    eng = AuroraCapEngine({"vit": vcfg, "llm": lcfg}, w, max_frames=4, max_batch=1, max_ctx=256, max_new_tokens=16)
    try:
        m = AuroraModel(eng, eos_token_id=None, slowfast=True)
        px = torch.randn(1, 4, 3, 56, 56, generator=torch.Generator().manual_seed(8)).half()
        ids = torch.tensor([[1, 17, -200, 18, -200, 19, -200, 20, -200, 21]])
        m.visual_encoder.reset_tome_r(0.5)
        out = m({"pixel_values": px, "input_ids": ids}, mode="inference")
        assert m.visual_encoder.visual_token_merge_ratio == 1.0               # the reference's side effect (aurora.py:231)
        feats = O.visual_features_slowfast(px[0].float(), w["vit"], w["projector"], vcfg, 0.5, q=O.fp16_storage)
        ref = O.splice_slowfast(ids[0], w["llm"]["embed_tokens.weight"], feats)
        emb = out["inputs_embeds"][0].float().cpu()
        n0, n = feats[0].shape[0], feats[1].shape[0]
        assert n0 == 16 and n < n0 and emb.shape == ref.shape == (6 + n0 + 3 * n, lcfg["hidden_size"])
        np.testing.assert_array_equal(emb[0].numpy(), w["llm"]["embed_tokens.weight"][1].numpy())     # text rows exact
        np.testing.assert_array_equal(emb[2 + n0].numpy(), w["llm"]["embed_tokens.weight"][18].numpy())
        assert rel_l2(emb[2:2 + n0], ref[2:2 + n0]) < 5e-3                     # frame 0: unmerged, row for row
        lo = 3 + n0
        assert rel_l2(emb[lo:lo + n].mean(0), ref[lo:lo + n].mean(0)) < 1e-2   # merged frames: order may differ at near ties
        cont = m.llm.generate(**out, do_sample=False, num_beams=1, max_new_tokens=6)
        assert cont.shape == (1, 6)
        # single image: slow-fast is bypassed (f == 1), ratio knob untouched
        m.visual_encoder.reset_tome_r(0.5)
        one = m({"pixel_values": px[:, 0], "input_ids": torch.tensor([[1, -200, 5]])}, mode="inference")
        assert one["inputs_embeds"].shape[1] == 2 + n and m.visual_encoder.visual_token_merge_ratio == 0.5
        # fewer frames than markers is an error in slow-fast mode (utils.py:361), not a silent drop
        m.visual_encoder.reset_tome_r(0.5)
        with pytest.raises(IndexError):
            m({"pixel_values": px[:, :3], "input_ids": ids}, mode="inference")
    finally:
        eng.close()


@pytest.mark.parametrize("h,w", [(98, 98), (70, 126), (84, 112), (56, 56)])
def test_inputs_larger_than_the_native_grid(h, w):
    """max_image decouples the workspace capacity from the checkpoint's position grid: a 56-pixel (4x4) tower encodes
    98x98 (7x7) or 70x126 (5x9) inputs through the up-sampled position table, like the reference does for any size."""
    from aurora_amd.engine import AuroraCapEngine
    cfg = dict(VCFG, image_size=56)
    wts = rand_vit_weights(cfg, 6)
    eng = AuroraCapEngine({"vit": cfg, "llm": None}, {"vit": wts}, max_frames=2, max_batch=1, max_ctx=128, max_new_tokens=8, max_image=126)
    try:
        px = torch.randn(2, 3, h, w, generator=torch.Generator().manual_seed(h * 3 + w)).half().float()
        out = eng.vit_encode(px, eng.tome_r(1.0, h, w)).float().cpu()
        ref = O.vit_features(px, wts, cfg, 1.0)
        assert out.shape == ref.shape == (2, (h // 14) * (w // 14), 64)
        assert rel_l2(out, ref) < 5e-3
        out = eng.vit_encode(px, eng.tome_r(0.4, h, w)).float().cpu()
        ref = O.vit_features(px, wts, cfg, 0.4, q=O.fp16_storage)
        # merged case (seeds chosen without a near-tie flip between fp16 storage and fp32: with seed h + w the two ORACLE
        # variants themselves diverge by 27 % on 70x126 - SURVEY 8c on why index equality is checked per kernel instead)
        assert out.shape == ref.shape and rel_l2(out.mean(1), ref.mean(1)) < 1e-2
        with pytest.raises(ValueError):
            eng.vit_encode(torch.zeros(1, 3, 140, 140), 0)                    # beyond max_image
    finally:
        eng.close()
