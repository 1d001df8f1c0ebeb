#!/usr/bin/env python3
"""Generate tests/golden/g10_preprocess.npz: outputs of the input stage the reference builds at inference.py:58-63,
71-72 - `CLIPImageProcessor(size=378, crop_size=378)(frames)['pixel_values'].to(float16)` - produced by the
transformers + Pillow builds installed in this container (third-party; versions recorded in the fixture), and the
frame-sampling table of load_video.py:38-44.

Fixtures are DATA ONLY.  Inputs are regenerated from (seed, recipe) by tests/util.py::preprocess_case_input, their
sha256 is stored to detect RNG drift; outputs are stored as sha256 + a few full rows (random images compress badly).
Re-run:  python tests/golden/make_golden_preprocess.py      (CPU only, ~20 s)
"""
import hashlib
import os
import sys
import warnings

import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, os.path.join(HERE, "..", ".."))

from tests.util import PREPROCESS_CASES, PREPROCESS_ROWS, preprocess_case_input  # noqa: E402


def sha(a: np.ndarray) -> str:
    return hashlib.sha256(np.ascontiguousarray(a).tobytes()).hexdigest()


def main():
    import PIL
    import torch
    import transformers
    from PIL import Image
    from transformers import CLIPImageProcessor
    warnings.filterwarnings("ignore")
    proc = CLIPImageProcessor(size=378, crop_size=378)           # the constructor arguments of inference.py:58-63
    out = {"versions": np.array([f"pillow={PIL.__version__}", f"transformers={transformers.__version__}",
                                 f"numpy={np.__version__}", f"processor={type(proc).__name__}"])}
    for name, (h, w, kind, seed) in PREPROCESS_CASES.items():
        img = preprocess_case_input(h, w, kind, seed)
        pv32 = proc(img, return_tensors="pt")["pixel_values"][0]
        pv16 = pv32.to(torch.float16).numpy()
        # the uint8 image after resize + crop, straight from Pillow (what the processor normalises)
        short, long = (w, h) if w <= h else (h, w)
        new_long = int(378 * long / short)
        nh, nw = (new_long, 378) if w <= h else (378, new_long)
        rz = np.asarray(Image.fromarray(img).resize((nw, nh), resample=Image.BICUBIC))
        top, left = (nh - 378) // 2, (nw - 378) // 2
        u8 = rz[top:top + 378, left:left + 378]
        out[f"{name}.in_sha"] = np.array(sha(img))
        out[f"{name}.u8_sha"] = np.array(sha(u8))
        out[f"{name}.u8_rows"] = u8[PREPROCESS_ROWS].copy()
        out[f"{name}.pv16_sha"] = np.array(sha(pv16))
        out[f"{name}.pv32_sha"] = np.array(sha(pv32.numpy()))
        out[f"{name}.pv16_rows"] = pv16[:, PREPROCESS_ROWS].copy()
        print(name, (h, w), "->", (nh, nw), out[f"{name}.pv16_sha"])
    # rescale + normalise as a function of the uint8 value, per channel (378x378 input: no resampling)
    g = np.tile(np.arange(256, dtype=np.uint8), 378 * 378 // 256 + 1)[:378 * 378].reshape(378, 378)
    pv = proc(np.stack([g, g, g], -1), return_tensors="pt")["pixel_values"][0].numpy()
    lut = np.stack([np.array([pv[c][g == v][0] for v in range(256)], np.float32) for c in range(3)])
    out["lut32"] = lut
    # load_video.py:38-44, evaluated literally
    pairs = [(100, 8), (100, 1), (8, 8), (5, 8), (1, 8), (2, 1), (17, 16), (300, 16), (9, 4), (1000, 32)]
    tab = []
    for total_frames, num_frm in pairs:
        sampled_frm = min(total_frames, num_frm)
        indices = np.linspace(0, total_frames - 1, sampled_frm, dtype=int)
        if total_frames - 1 not in indices:
            indices = np.append(indices, total_frames - 1)
        tab.append(indices.astype(np.int64))
    out["sampling.pairs"] = np.array(pairs, np.int64)
    for i, t in enumerate(tab):
        out[f"sampling.{i}"] = t
    path = os.path.join(HERE, "g10_preprocess.npz")
    np.savez_compressed(path, **out)
    print("wrote", path, os.path.getsize(path), "bytes")


if __name__ == "__main__":
    main()
