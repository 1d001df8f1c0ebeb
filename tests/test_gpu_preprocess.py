"""GPU: the HIP input stage (aur_preprocess_frames via aurora_amd.preprocess.FramePreprocessor) must be BIT-EXACT
against the oracle and against the fixtures written from the reference's own processor call."""
import hashlib

import numpy as np
import pytest
import torch

from oracle import preprocess_ref as R
from tests.util import PREPROCESS_CASES, PREPROCESS_ROWS, golden, preprocess_case_input

pytestmark = pytest.mark.gpu


def sha(a):
    return hashlib.sha256(np.ascontiguousarray(a).tobytes()).hexdigest()


@pytest.fixture(scope="module")
def pre():
    from aurora_amd.preprocess import FramePreprocessor
    return FramePreprocessor()


@pytest.mark.parametrize("name", list(PREPROCESS_CASES))
def test_matches_reference_processor_fixture(pre, name):
    g10 = golden("g10_preprocess.npz")
    h, w, kind, seed = PREPROCESS_CASES[name]
    img = preprocess_case_input(h, w, kind, seed)
    assert sha(img) == str(g10[f"{name}.in_sha"])
    pv = pre(torch.from_numpy(img)[None].cuda())[0].cpu().numpy()
    assert np.array_equal(pv[:, PREPROCESS_ROWS].view(np.uint16), g10[f"{name}.pv16_rows"].view(np.uint16))
    assert sha(pv) == str(g10[f"{name}.pv16_sha"])


@pytest.mark.parametrize("h,w,f", [(480, 640, 8), (720, 1280, 8), (1080, 1920, 3), (640, 360, 5), (33, 1000, 2), (378, 378, 4),
                                   (2160, 3840, 1), (100, 100, 16)])
def test_batch_bit_exact_against_oracle(pre, h, w, f):
    rng = np.random.default_rng(h * 7 + w)
    frames = rng.integers(0, 256, (f, h, w, 3), dtype=np.uint8)
    got = pre(torch.from_numpy(frames).cuda()).cpu().numpy()
    want = R.clip_preprocess(frames)
    assert got.shape == (f, 3, 378, 378) and got.dtype == np.float16
    assert np.array_equal(got.view(np.uint16), want.view(np.uint16))


def test_plan_cache_other_normalisation_and_errors(pre):
    from aurora_amd.preprocess import FramePreprocessor
    rng = np.random.default_rng(5)
    a = rng.integers(0, 256, (2, 240, 320, 3), dtype=np.uint8)
    b = rng.integers(0, 256, (2, 320, 240, 3), dtype=np.uint8)
    for _ in range(2):                                               # alternating sizes reuse the cached plans
        for fr in (a, b):
            got = pre(torch.from_numpy(fr).cuda()).cpu().numpy()
            assert np.array_equal(got.view(np.uint16), R.clip_preprocess(fr).view(np.uint16))
    m, s = (0.5, 0.5, 0.5), (0.5, 0.25, 0.125)
    p2 = FramePreprocessor(mean=m, std=s)
    got = p2(torch.from_numpy(a).cuda()).cpu().numpy()
    assert np.array_equal(got.view(np.uint16), R.clip_preprocess(a, mean=m, std=s).view(np.uint16))
    with pytest.raises(ValueError):
        pre(torch.zeros(2, 10, 10, 3))                               # not on the device / wrong dtype
    with pytest.raises(ValueError):
        pre(torch.zeros(2, 10, 10, 4, dtype=torch.uint8, device="cuda"))


def test_feeds_vit_encode():
    """Decoded frames -> HIP input stage -> aur_vit_encode, against the oracle chain on the same frames."""
    from aurora_amd.engine import AuroraCapEngine
    from aurora_amd.preprocess import FramePreprocessor
    from oracle import aurora_oracle as O
    from tests.util import rand_vit_weights, rel_l2
    cfg = dict(hidden_size=64, num_attention_heads=4, num_hidden_layers=4, intermediate_size=128, patch_size=14, image_size=56,
               num_channels=3, hidden_act="quick_gelu", layer_norm_eps=1e-5)
    w = rand_vit_weights(cfg, 3)
    rng = np.random.default_rng(9)
    frames = rng.integers(0, 256, (2, 90, 120, 3), dtype=np.uint8)
    pre = FramePreprocessor(image=56)
    px = pre(torch.from_numpy(frames).cuda())
    want_px = R.clip_preprocess(frames, size=56)
    assert np.array_equal(px.cpu().numpy().view(np.uint16), want_px.view(np.uint16))
    eng = AuroraCapEngine({"vit": cfg, "llm": None}, {"vit": w}, max_frames=2, max_batch=1, max_ctx=128, max_new_tokens=8)
    try:
        out = eng.vit_encode(px, eng.tome_r(1.0)).float().cpu()
        ref = O.vit_features(torch.from_numpy(want_px).float(), w, cfg, 1.0)
        assert out.shape == ref.shape
        assert rel_l2(out, ref) < 5e-3                               # fp16 storage / fp32 accumulate vs fp32 oracle chain
    finally:
        eng.close()
