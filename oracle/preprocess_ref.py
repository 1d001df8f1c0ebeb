"""TEST INFRASTRUCTURE ONLY - CPU restatement of the reference's input stage.

Only tests/, __graft_entry__.smoke() and bench.py's cpu_baseline leg may import this module; the product path
(aurora_amd/) never does.

What the reference does between a decoded clip and `data["pixel_values"]`:

  * frame sampling            src/xtuner/xtuner/tools/load_video.py:36-44 (np.linspace + "append the last frame")
  * CLIPImageProcessor(size=378, crop_size=378)(frames)["pixel_values"].to(float16)     inference.py:58-63, 71-75
      - resize shortest edge to 378, PIL BICUBIC        (third party: transformers image_transforms.resize ->
                                                          PIL.Image.resize -> Pillow src/libImaging/Resample.c)
      - centre crop 378x378                             (transformers image_transforms.center_crop)
      - rescale by 1/255, normalise by the CLIP mean/std (transformers image_transforms.rescale / normalize)

Pillow and transformers are third-party dependencies that are not vendored under /root/reference; their
published algorithms are restated here (Pillow 8-bit resampling: separable, horizontal pass first, uint8
intermediate, 22-bit fixed-point coefficients computed in doubles) and PINNED against the Pillow 12.2.0 /
transformers builds in this image by tests/test_preprocess.py and the fixtures written by
tests/golden/make_golden_preprocess.py.
"""
from __future__ import annotations

import math

import numpy as np

PRECISION_BITS = 32 - 8 - 2            # Resample.c: coefficients are int32 with 22 fractional bits
CLIP_MEAN = (0.48145466, 0.4578275, 0.40821073)
CLIP_STD = (0.26862954, 0.26130258, 0.27577711)


def sample_frame_indices(total_frames: int, num_frm: int) -> list[int]:
    """load_video.py:38-44: linspace(0, total-1, min(total, num_frm), dtype=int), then the last frame is appended
    if it is not already there (so num_frm=1 on a clip longer than one frame yields TWO frames)."""
    sampled = min(total_frames, num_frm)
    idx = np.linspace(0, total_frames - 1, sampled, dtype=int)
    if total_frames - 1 not in idx:
        idx = np.append(idx, total_frames - 1)
    return [int(i) for i in idx]


def resize_output_size(h: int, w: int, shortest_edge: int) -> tuple[int, int]:
    """transformers get_resize_output_image_size(default_to_square=False): the short side becomes `shortest_edge`,
    the long side int(shortest_edge * long / short)."""
    short, long = (w, h) if w <= h else (h, w)
    new_short, new_long = shortest_edge, int(shortest_edge * long / short)
    return (new_long, new_short) if w <= h else (new_short, new_long)


def _bicubic(x: float) -> float:
    a = -0.5
    if x < 0.0:
        x = -x
    if x < 1.0:
        return ((a + 2.0) * x - (a + 3.0)) * x * x + 1
    if x < 2.0:
        return (((x - 5) * x + 8) * x - 4) * a
    return 0.0


def precompute_coeffs(in_size: int, out_size: int):
    """Resample.c precompute_coeffs + normalize_coeffs_8bpc for the bicubic filter (support 2) over the full axis.
    Returns (ksize, bounds[out,2] = (first input index, tap count), coeffs[out,ksize] int32)."""
    scale = float(in_size) / out_size
    filterscale = max(scale, 1.0)
    support = 2.0 * filterscale
    ksize = int(math.ceil(support)) * 2 + 1
    bounds = np.zeros((out_size, 2), np.int32)
    kk = np.zeros((out_size, ksize), np.int32)
    ss = 1.0 / filterscale
    for xx in range(out_size):
        center = (xx + 0.5) * scale
        xmin = int(center - support + 0.5)
        if xmin < 0:
            xmin = 0
        xmax = int(center + support + 0.5)
        if xmax > in_size:
            xmax = in_size
        xmax -= xmin
        k = [_bicubic((x + xmin - center + 0.5) * ss) for x in range(xmax)]
        ww = 0.0
        for w in k:
            ww += w
        for x in range(xmax):
            v = k[x] / ww if ww != 0.0 else k[x]
            kk[xx, x] = int(-0.5 + v * (1 << PRECISION_BITS)) if v < 0 else int(0.5 + v * (1 << PRECISION_BITS))
        bounds[xx] = (xmin, xmax)
    return ksize, bounds, kk


def _resample_axis(img: np.ndarray, out_size: int, axis: int) -> np.ndarray:
    """One 8-bit pass along `axis` of an [H, W, C] uint8 image: sum(pixel * k) + 2^21, >> 22, clip to 0..255."""
    in_size = img.shape[axis]
    _, bounds, kk = precompute_coeffs(in_size, out_size)
    src = np.moveaxis(img, axis, 0).astype(np.int64)
    out = np.empty((out_size,) + src.shape[1:], np.uint8)
    for xx in range(out_size):
        x0, n = bounds[xx]
        acc = np.tensordot(kk[xx, :n].astype(np.int64), src[x0:x0 + n], axes=(0, 0)) + (1 << (PRECISION_BITS - 1))
        out[xx] = np.clip(acc >> PRECISION_BITS, 0, 255).astype(np.uint8)
    return np.moveaxis(out, 0, axis)


def pil_bicubic_resize(img: np.ndarray, out_h: int, out_w: int) -> np.ndarray:
    """ImagingResample: horizontal pass (skipped when the width is unchanged), then vertical pass (same)."""
    assert img.dtype == np.uint8 and img.ndim == 3
    if img.shape[1] != out_w:
        img = _resample_axis(img, out_w, 1)
    if img.shape[0] != out_h:
        img = _resample_axis(img, out_h, 0)
    return img


def center_crop(img: np.ndarray, ch: int, cw: int) -> np.ndarray:
    h, w = img.shape[:2]
    top, left = (h - ch) // 2, (w - cw) // 2
    assert top >= 0 and left >= 0                      # always true after a shortest-edge resize to the crop size
    return img[top:top + ch, left:left + cw]


def normalise_lut(mean=CLIP_MEAN, std=CLIP_STD, rescale: float = 1 / 255) -> np.ndarray:
    """[3, 256] float16: rescale (uint8 * python float -> float64 -> float32), normalise in float32, then the
    reference's .to(torch.float16) (inference.py:72)."""
    v = (np.arange(256, dtype=np.uint8) * rescale).astype(np.float32)
    m = np.array(mean, dtype=np.float32)[:, None]
    s = np.array(std, dtype=np.float32)[:, None]
    return ((v[None, :] - m) / s).astype(np.float32).astype(np.float16)


def clip_preprocess_u8(frame: np.ndarray, size: int = 378) -> np.ndarray:
    """[H, W, 3] uint8 -> [size, size, 3] uint8 after resize + centre crop."""
    nh, nw = resize_output_size(frame.shape[0], frame.shape[1], size)
    return center_crop(pil_bicubic_resize(frame, nh, nw), size, size)


def clip_preprocess(frames: np.ndarray, size: int = 378, mean=CLIP_MEAN, std=CLIP_STD) -> np.ndarray:
    """[f, H, W, 3] uint8 -> [f, 3, size, size] float16 pixel_values."""
    lut = normalise_lut(mean, std)
    out = np.empty((len(frames), 3, size, size), np.float16)
    for i, fr in enumerate(frames):
        u8 = clip_preprocess_u8(fr, size)
        for c in range(3):
            out[i, c] = lut[c][u8[:, :, c]]
    return out
