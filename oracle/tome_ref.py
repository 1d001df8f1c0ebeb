"""ctypes binding of oracle/tome_ref.c (bit-exact ToMe oracle).  TEST INFRASTRUCTURE ONLY."""
from __future__ import annotations

import ctypes
import os
import subprocess

import numpy as np

_HERE = os.path.dirname(os.path.abspath(__file__))
_SO = os.path.join(_HERE, "_build", "libtome_ref.so")


def build(force: bool = False) -> str:
    src = os.path.join(_HERE, "tome_ref.c")
    if force or not os.path.exists(_SO) or os.path.getmtime(_SO) < os.path.getmtime(src):
        subprocess.check_call(["make", "-C", _HERE, "-B" if force else "-s"])
    return _SO


_lib = None


def lib():
    global _lib
    if _lib is None:
        _lib = ctypes.CDLL(build())
        fp = ctypes.POINTER(ctypes.c_float)
        ip = ctypes.POINTER(ctypes.c_int32)
        _lib.tome_match_ref.argtypes = [fp, ctypes.c_int, ctypes.c_int, ctypes.c_int, fp, ip, ip, ip, ip]
        _lib.tome_match_ref.restype = ctypes.c_int
        _lib.tome_merge_ref.argtypes = [fp, fp, ctypes.c_int, ctypes.c_int, ctypes.c_int, ip, ip, ip, fp, fp]
        _lib.tome_merge_ref.restype = ctypes.c_int
    return _lib


def _fp(a):
    return a.ctypes.data_as(ctypes.POINTER(ctypes.c_float))


def _ip(a):
    return a.ctypes.data_as(ctypes.POINTER(ctypes.c_int32))


def clamp_r(t: int, r: int) -> int:
    return max(0, min(r, (t - 1) // 2))


def match(metric: np.ndarray, r: int):
    """metric [f, t, c] float32 -> dict of int32/float32 arrays (per frame).  r is clamped."""
    metric = np.ascontiguousarray(metric, dtype=np.float32)
    f, t, c = metric.shape
    r = clamp_r(t, r)
    if r <= 0:
        return None
    ta = (t + 1) // 2
    out = dict(r=r,
               node_max=np.empty((f, ta), np.float32), node_idx=np.empty((f, ta), np.int32),
               unm_idx=np.empty((f, ta - r), np.int32), src_idx=np.empty((f, r), np.int32),
               dst_idx=np.empty((f, r), np.int32))
    for i in range(f):
        rc = lib().tome_match_ref(_fp(metric[i]), t, c, r, _fp(out["node_max"][i]), _ip(out["node_idx"][i]),
                                  _ip(out["unm_idx"][i]), _ip(out["src_idx"][i]), _ip(out["dst_idx"][i]))
        assert rc == 0
    return out


def merge(x: np.ndarray, size: np.ndarray, m) -> tuple:
    """x [f, t, d] float32, size [f, t] float32, m from match() -> (x' [f, t-r, d], size' [f, t-r])."""
    x = np.ascontiguousarray(x, dtype=np.float32)
    size = np.ascontiguousarray(size, dtype=np.float32)
    if m is None:
        return x.copy(), size.copy()
    f, t, d = x.shape
    r = m["r"]
    xo = np.empty((f, t - r, d), np.float32)
    so = np.empty((f, t - r), np.float32)
    for i in range(f):
        rc = lib().tome_merge_ref(_fp(x[i]), _fp(size[i]), t, d, r, _ip(m["unm_idx"][i]), _ip(m["src_idx"][i]),
                                  _ip(m["dst_idx"][i]), _fp(xo[i]), _fp(so[i]))
        assert rc == 0
    return xo, so
