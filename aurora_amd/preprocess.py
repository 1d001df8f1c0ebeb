"""Input stage on the GPU: decoded rgb24 frames -> `pixel_values`.

Host side of aur_preprocess_* (include/aurora_hip.h).  Replaces, for frames that are already decoded,

    image_processor = CLIPImageProcessor.from_pretrained(..., size=378, crop_size=378)          # inference.py:58-63
    image_tensor = image_processor(video_frames, return_tensors='pt')['pixel_values']           # inference.py:71
    image_tensor = [_image.to(dtype=torch.float16).cuda() for _image in image_tensor]           # inference.py:72

with two HIP kernels whose output is bit-identical to that processor (Pillow's 8-bit bicubic is integer arithmetic
once the taps are known; the taps are computed in doubles on the host exactly as Pillow does).  Frame *decoding*
(pyav / decord) stays on the host; the sampling rule of load_video.py:36-44 is `sample_frame_indices`.
"""
from __future__ import annotations

import ctypes as C
import json
import os
from typing import Sequence

import numpy as np
import torch

from ._lib import AuroraHipError, lib

CLIP_MEAN = (0.48145466, 0.4578275, 0.40821073)      # OPENAI_CLIP_MEAN / _STD: the defaults of CLIPImageProcessor
CLIP_STD = (0.26862954, 0.26130258, 0.27577711)


def sample_frame_indices(total_frames: int, num_frm: int) -> list[int]:
    """load_video.py:38-44: np.linspace(0, total-1, min(total, num_frm), dtype=int) plus the last frame when it is
    missing - `--num_frm 1` on a longer clip therefore yields two frames."""
    if total_frames < 1 or num_frm < 1:
        raise ValueError(f"total_frames={total_frames}, num_frm={num_frm}")
    idx = np.linspace(0, total_frames - 1, min(total_frames, num_frm), dtype=int).tolist()
    if total_frames - 1 not in idx:
        idx.append(total_frames - 1)
    return [int(i) for i in idx]


def read_video_pyav(path: str, num_frm: int = 8):
    """Uniform frame sampling with the contract of src/xtuner/xtuner/tools/load_video.py:31-71
    (aurora_amd.preprocess.sample_frame_indices: linspace over min(total, num_frm) frames + the last frame);
    webm / mkv and streams without a frame count are decoded packet by packet first, as the reference does."""
    try:
        import av
    except ImportError as e:
        raise RuntimeError("PyAV (`av`) is required to decode video input") from e
    def by_packets():
        """load_video.py:21-28, 48-56: decode every frame, then sample - webm / mkv, streams without a frame count, and the
        reference's bare `except:` fall-back when the stream path fails on an mp4 (bad index, truncated file)."""
        frames = [f for f in av.open(path).decode(video=0)]
        return np.stack([frames[i].to_ndarray(format="rgb24") for i in sample_frame_indices(len(frames), num_frm)])

    if "webm" in path or "mkv" in path:
        return by_packets()
    try:                                                           # load_video.py:35-47: "for mp4, we try loading with stream first"
        container = av.open(path)
        total = container.streams.video[0].frames
        if total <= 0:                                             # no frame count in the header: the reference raises (np.stack of nothing); deliberate fall-back
            return by_packets()
        idx = sample_frame_indices(total, num_frm)
        want, out = set(idx), {}
        for i, frame in enumerate(container.decode(video=0)):
            if i in want:
                out[i] = frame.to_ndarray(format="rgb24")
            if i >= idx[-1]:
                break
        # a header that promised more frames than the stream holds: the reference returns the (shorter) list of frames it found
        # (record_video_length_stream, load_video.py:7-16, 46) - no re-sampling.  Only when NOTHING was found, where the reference
        # dies in np.stack([]), do we decode by packets instead (deliberate: a caption beats a ValueError)
        if not out:
            return by_packets()
        return np.stack([out[i] for i in idx if i in out])
    except Exception:                                              # noqa: BLE001 - the reference catches everything here too
        return by_packets()


def host_plan(in_h: int, in_w: int, image: int = 378) -> np.ndarray:
    """int32 resampling plan for one input size (geometry header + per-column / per-row taps); host only."""
    L = lib()
    n = L.aur_preprocess_plan_len(in_h, in_w, image)
    if n < 0:
        raise AuroraHipError(f"unsupported frame size {in_h}x{in_w} -> {image}")
    plan = np.empty(n, np.int32)
    rc = L.aur_preprocess_plan(in_h, in_w, image, plan.ctypes.data_as(C.c_void_p), n)
    if rc != 0:
        raise AuroraHipError(f"aur_preprocess_plan failed (status {rc})")
    return plan


def normalise_lut(mean: Sequence[float] = CLIP_MEAN, std: Sequence[float] = CLIP_STD, rescale: float = 1 / 255) -> np.ndarray:
    """[3, 256] fp16 = float16(float32((float32(v * rescale) - mean) / std)): the processor's rescale (float64 product
    rounded to float32) and normalise (float32) followed by the reference's cast to half."""
    v = (np.arange(256, dtype=np.float64) * rescale).astype(np.float32)
    m = np.asarray(mean, np.float32)[:, None]
    s = np.asarray(std, np.float32)[:, None]
    return ((v[None] - m) / s).astype(np.float16)


class FramePreprocessor:
    """`pixel_values = FramePreprocessor()(frames_u8)` with frames_u8 a CUDA uint8 tensor [f, H, W, 3] (rgb24)."""

    def __init__(self, image: int = 378, mean: Sequence[float] = CLIP_MEAN, std: Sequence[float] = CLIP_STD,
                 rescale: float = 1 / 255, device: str | torch.device = "cuda"):
        self.image = int(image)
        self.device = torch.device(device)
        self._lib = lib()
        self._lut = torch.from_numpy(normalise_lut(mean, std, rescale).view(np.int16)).to(self.device)
        self._plans: dict[tuple[int, int], torch.Tensor] = {}
        self._tmp: torch.Tensor | None = None

    @classmethod
    def from_config(cls, path: str, **kw) -> "FramePreprocessor":
        """Read a HF preprocessor_config.json (image_mean / image_std / rescale_factor / crop_size)."""
        with open(os.path.join(path, "preprocessor_config.json") if os.path.isdir(path) else path) as f:
            c = json.load(f)
        crop = c.get("crop_size", 378)
        crop = crop.get("height", 378) if isinstance(crop, dict) else crop
        return cls(image=kw.pop("image", crop), mean=c.get("image_mean", CLIP_MEAN), std=c.get("image_std", CLIP_STD),
                   rescale=c.get("rescale_factor", 1 / 255), **kw)

    def plan(self, in_h: int, in_w: int) -> torch.Tensor:
        key = (int(in_h), int(in_w))
        if key not in self._plans:
            self._plans[key] = torch.from_numpy(host_plan(in_h, in_w, self.image)).to(self.device)
        return self._plans[key]

    def __call__(self, frames: torch.Tensor, out: torch.Tensor | None = None) -> torch.Tensor:
        if frames.dtype != torch.uint8 or frames.dim() != 4 or frames.shape[-1] != 3 or not frames.is_cuda:
            raise ValueError("frames must be a CUDA uint8 tensor [f, H, W, 3]")
        frames = frames.contiguous()
        f, h, w, _ = frames.shape
        plan = self.plan(h, w)
        need = self._lib.aur_preprocess_tmp_bytes(f, h, w, self.image)
        if self._tmp is None or self._tmp.numel() < need:
            self._tmp = torch.empty(need, dtype=torch.uint8, device=self.device)
        if out is None:
            out = torch.empty(f, 3, self.image, self.image, dtype=torch.float16, device=self.device)
        rc = self._lib.aur_preprocess_frames(frames.data_ptr(), f, h, w, self.image, plan.data_ptr(), self._lut.data_ptr(),
                                             self._tmp.data_ptr(), out.data_ptr(), torch.cuda.current_stream().cuda_stream)
        if rc != 0:
            raise AuroraHipError(f"aur_preprocess_frames failed (status {rc})")
        return out
