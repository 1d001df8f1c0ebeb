"""CPU: the input-stage oracle against the fixtures produced by the reference's own processor call
(tests/golden/make_golden_preprocess.py), and the host half of the product path (resampling plan computed by
libaurora_hip.so, normalise LUT, frame sampling) against the oracle.  No device work here."""
import hashlib

import numpy as np
import pytest

from oracle import preprocess_ref as R
from tests.util import PREPROCESS_CASES, PREPROCESS_ROWS, golden, preprocess_case_input


def sha(a):
    return hashlib.sha256(np.ascontiguousarray(a).tobytes()).hexdigest()


@pytest.fixture(scope="module")
def g10():
    return golden("g10_preprocess.npz")


@pytest.mark.parametrize("name", list(PREPROCESS_CASES))
def test_oracle_matches_reference_processor(g10, name):
    h, w, kind, seed = PREPROCESS_CASES[name]
    img = preprocess_case_input(h, w, kind, seed)
    assert sha(img) == str(g10[f"{name}.in_sha"]), "test-input RNG drift: regenerate the fixture"
    u8 = R.clip_preprocess_u8(img)
    assert np.array_equal(u8[PREPROCESS_ROWS], g10[f"{name}.u8_rows"])
    assert sha(u8) == str(g10[f"{name}.u8_sha"])
    pv = R.clip_preprocess(img[None])[0]
    assert np.array_equal(pv[:, PREPROCESS_ROWS].view(np.uint16), g10[f"{name}.pv16_rows"].view(np.uint16))
    assert sha(pv) == str(g10[f"{name}.pv16_sha"])


def test_oracle_lut_matches_processor_fp32_and_fp16(g10):
    lut32 = g10["lut32"]
    v = (np.arange(256, dtype=np.uint8) * (1 / 255)).astype(np.float32)
    mine32 = (v[None] - np.array(R.CLIP_MEAN, np.float32)[:, None]) / np.array(R.CLIP_STD, np.float32)[:, None]
    assert np.array_equal(mine32, lut32)
    assert np.array_equal(R.normalise_lut().view(np.uint16), lut32.astype(np.float16).view(np.uint16))


def test_oracle_against_live_pillow():
    Image = pytest.importorskip("PIL.Image")
    rng = np.random.default_rng(11)
    for h, w in [(33, 1000), (1000, 33), (378, 379), (500, 500), (2160, 3840)]:
        img = rng.integers(0, 256, (h, w, 3), dtype=np.uint8)
        nh, nw = R.resize_output_size(h, w, 378)
        ref = np.asarray(Image.fromarray(img).resize((nw, nh), resample=Image.BICUBIC))
        assert np.array_equal(R.pil_bicubic_resize(img, nh, nw), ref), (h, w)


def test_sampling_table(g10):
    from aurora_amd.preprocess import sample_frame_indices
    pairs = g10["sampling.pairs"]
    for i, (total, num) in enumerate(pairs):
        want = g10[f"sampling.{i}"].tolist()
        assert R.sample_frame_indices(int(total), int(num)) == want
        assert sample_frame_indices(int(total), int(num)) == want
    assert sample_frame_indices(100, 1) == [0, 99]                    # --num_frm 1 on a video yields two frames
    with pytest.raises(ValueError):
        sample_frame_indices(0, 8)


@pytest.mark.parametrize("h,w", [(480, 640), (720, 1280), (640, 480), (378, 378), (200, 300), (97, 131), (1080, 1920),
                                 (379, 377), (1000, 378), (2160, 3840), (33, 1000)])
def test_library_plan_equals_oracle_taps(h, w):
    """aur_preprocess_plan (host C++, doubles) must reproduce Pillow's taps exactly: geometry, first index, count and
    every int32 coefficient for the 378 columns / rows that survive the centre crop."""
    from aurora_amd.preprocess import host_plan
    plan = host_plan(h, w, 378)
    nh, nw = R.resize_output_size(h, w, 378)
    top, left = (nh - 378) // 2, (nw - 378) // 2
    ks_h, bh, kh = R.precompute_coeffs(w, nw)
    ks_v, bv, kv = R.precompute_coeffs(h, nh)
    assert plan[:8].tolist() == [nh, nw, top, left, ks_h, ks_v, h, w]
    o = 8
    assert np.array_equal(plan[o:o + 756].reshape(378, 2), bh[left:left + 378]); o += 756
    assert np.array_equal(plan[o:o + 378 * ks_h].reshape(ks_h, 378).T, kh[left:left + 378]); o += 378 * ks_h   # tap-major
    assert np.array_equal(plan[o:o + 756].reshape(378, 2), bv[top:top + 378]); o += 756
    assert np.array_equal(plan[o:o + 378 * ks_v].reshape(378, ks_v), kv[top:top + 378]); o += 378 * ks_v
    assert o == plan.size


def test_plan_rejects_bad_sizes_and_lut_matches_oracle():
    from aurora_amd._lib import AuroraHipError
    from aurora_amd.preprocess import host_plan, normalise_lut
    with pytest.raises(AuroraHipError):
        host_plan(0, 10)
    with pytest.raises(AuroraHipError):
        host_plan(10, 100000)
    with pytest.raises(AuroraHipError):
        host_plan(378 * 40, 378 * 40)                                # > 32x reduction: taps would not fit
    assert np.array_equal(normalise_lut().view(np.uint16), R.normalise_lut().view(np.uint16))
    m, s = (0.5, 0.4, 0.3), (0.2, 0.25, 0.3)
    assert np.array_equal(normalise_lut(m, s).view(np.uint16), R.normalise_lut(m, s).view(np.uint16))


def test_read_video_pyav_falls_back_to_packet_decode(monkeypatch):
    """load_video.py:35-56: mp4s go through the stream path first; ANY failure there (the reference uses a bare `except:`) and
    every webm / mkv decode all frames and sample afterwards.  A fake `av` stands in for PyAV (not installed here)."""
    import sys
    import types
    from aurora_amd import preprocess as P

    class Frame:
        def __init__(self, i):
            self.i = i

        def to_ndarray(self, format):
            return np.full((2, 2, 3), self.i, np.uint8)

    class Container:
        def __init__(self, nframes, header, broken_stream):
            self.nframes, self.broken = nframes, broken_stream
            self.streams = types.SimpleNamespace(video=[types.SimpleNamespace(frames=header)])
            self.decodes = 0

        def decode(self, video=0):
            self.decodes += 1
            if self.broken and self.decodes == 1:
                raise OSError("corrupt index")
            return iter([Frame(i) for i in range(self.nframes)])

    opened = []

    def make_av(nframes, header, broken=False):
        def open_(path):
            c = Container(nframes, header, broken and not opened)
            opened.append(c)
            return c
        return types.SimpleNamespace(open=open_)

    def frames_of(path, nframes, header, broken=False, num_frm=4):
        opened.clear()
        monkeypatch.setitem(sys.modules, "av", make_av(nframes, header, broken))
        return [int(f[0, 0, 0]) for f in P.read_video_pyav(path, num_frm)]

    want = P.sample_frame_indices(20, 4)                          # [0, 6, 12, 19]
    assert frames_of("clip.mp4", 20, 20) == want                  # stream path
    assert frames_of("clip.mp4", 20, 20, broken=True) == want     # stream path raises -> packet decode (second open)
    assert frames_of("clip.mp4", 20, 0) == want                   # no frame count in the header
    # header promises more frames than exist: the reference returns the frames its stream walk found (indices 12 and 19 never turn
    # up among 12 frames) - record_video_length_stream, load_video.py:7-16 - without re-sampling
    assert frames_of("clip.mp4", 12, 20) == [0, 6]
    assert frames_of("clip.webm", 20, 20) == want and frames_of("clip.mkv", 20, 20) == want
