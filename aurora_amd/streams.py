"""HIP streams restricted to a subset of the compute units (`hipExtStreamCreateWithCUMask`).

Why: the decode step is HBM-bound and the front end (ViT + prefill GEMMs) is MFMA-bound.  Measured on MI355X
(`tools/cumask/cumask_probe.hip`): a streaming read reaches 7.13 TB/s on all 256 CUs and still 6.25 TB/s on 128 of them,
so an HBM-bound kernel gives up little when it leaves half of the CUs to a concurrent MFMA-bound kernel.

How the mask maps (same probe): the driver deals the mask bits round-robin over the 8 XCDs - bit i is CU (i // 8) of XCD
(i % 8) - and an XCD whose share of the mask is empty gets ALL its CUs back (workgroups are dealt to every XCD whatever the
mask says).  A stream can therefore be restricted to "k CUs of every XCD", never to whole XCDs.
"""
import ctypes as C
import os

import torch

_hip = None


def _runtime():
    global _hip
    if _hip is None:
        _hip = C.CDLL(os.path.join(os.path.dirname(torch.__file__), "lib", "libamdhip64.so"))   # torch's copy: one runtime per process
        _hip.hipExtStreamCreateWithCUMask.restype = C.c_int
        _hip.hipExtStreamCreateWithCUMask.argtypes = [C.POINTER(C.c_void_p), C.c_uint32, C.POINTER(C.c_uint32)]
        _hip.hipStreamDestroy.restype = C.c_int
        _hip.hipStreamDestroy.argtypes = [C.c_void_p]
    return _hip


def cu_mask_words(cus_per_xcd: int, from_top: bool = False, xcds: int = 8, cus_in_xcd: int = 32):
    """Mask words enabling `cus_per_xcd` CUs of every XCD: the lowest-numbered ones, or the highest with from_top."""
    if not 1 <= cus_per_xcd <= cus_in_xcd:
        raise ValueError(f"cus_per_xcd must be in [1, {cus_in_xcd}], got {cus_per_xcd}")
    nbits = xcds * cus_in_xcd
    words = [0] * ((nbits + 31) // 32)
    for i in range(nbits):
        cu = i // xcds
        if (cu >= cus_in_xcd - cus_per_xcd) if from_top else (cu < cus_per_xcd):
            words[i >> 5] |= 1 << (i & 31)
    return words


_CACHE = {}


def shared_cu_masked_stream(cus_per_xcd: int, from_top: bool = False, device=None) -> "torch.cuda.ExternalStream":
    """One CU-masked stream per (device, CUs, end) for the life of the process - what engines use.  torch's caching allocator
    remembers every stream a tensor was `record_stream`-ed on and records an event there when the tensor is freed, possibly long
    after the engine that ran on the stream is gone: a stream that tensors have been recorded on must never be destroyed (a
    destroyed one made the interpreter segfault when the clips of a finished caption_stream were garbage-collected)."""
    dev = torch.device("cuda", torch.cuda.current_device()) if device is None else torch.device(device)
    if dev.index is None:
        dev = torch.device("cuda", torch.cuda.current_device())
    key = (dev.index, int(cus_per_xcd), bool(from_top))
    if key not in _CACHE:
        _CACHE[key] = cu_masked_stream(cus_per_xcd, from_top, dev)
    return _CACHE[key]


def destroy_stream(stream: "torch.cuda.ExternalStream") -> None:
    """Release a stream made by `cu_masked_stream` (the wrapper does not own it).  The caller has synchronised it AND no tensor
    was ever `record_stream`-ed on it (see shared_cu_masked_stream)."""
    _runtime().hipStreamDestroy(C.c_void_p(stream.cuda_stream))


def cu_masked_stream(cus_per_xcd: int, from_top: bool = False, device=None) -> "torch.cuda.ExternalStream":
    """A new HIP stream whose kernels run on `cus_per_xcd` CUs of every XCD only (release it with `destroy_stream`)."""
    dev = torch.device("cuda", torch.cuda.current_device()) if device is None else torch.device(device)
    words = cu_mask_words(cus_per_xcd, from_top)
    arr = (C.c_uint32 * len(words))(*words)
    st = C.c_void_p()
    with torch.cuda.device(dev):
        rc = _runtime().hipExtStreamCreateWithCUMask(C.byref(st), len(words), arr)
    if rc != 0 or not st.value:
        raise RuntimeError(f"hipExtStreamCreateWithCUMask failed with status {rc}")
    return torch.cuda.ExternalStream(st.value, device=dev)


def overlap_pays(slots: int, mean_ctx: float, llm: dict) -> bool:
    """Does a front end beside the decode (CU-masked streams) pay for a decode of `slots` sequences at a mean context of `mean_ctx` tokens?
    Only while the decode step is dominated by the K / V stream, which half of the CUs can pull: a step of few slots is the weight stream
    through projections that want every CU, and masking it costs more than the overlap returns (measured with the 7B dims, bench.py:
    8 slots 5.4 captions/s overlapped against 5.8 with the front ends between the chunks; 96 / 128 slots 5.8 against 5.3 and 13.6
    against 11.4)."""
    L, h, m, V = llm["num_hidden_layers"], llm["hidden_size"], llm["intermediate_size"], llm["vocab_size"]
    kv_step = slots * mean_ctx * 4 * L * h                    # fp16 K + V of every cached token
    w_step = 2 * (L * (4 * h * h + 3 * h * m) + V * h)        # fp16 layer weights + lm_head
    return kv_step >= w_step
