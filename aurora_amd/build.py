"""Build libaurora_hip.so in-tree with hipcc for gfx950 (cross-compiles without a GPU).

    python -m aurora_amd.build [--force] [--sanitize | --ubsan | --device-asan]

The shared library has a plain C ABI (include/aurora_hip.h) and no torch / python dependency.
"""
from __future__ import annotations

import os
import subprocess
import sys
from concurrent.futures import ThreadPoolExecutor

HERE = os.path.dirname(os.path.abspath(__file__))
CSRC = os.path.join(HERE, "csrc")
OUT = os.path.join(HERE, "libaurora_hip.so")
SOURCES = ["engine.hip", "gemm.hip", "gemm256.hip", "attn.hip", "norm.hip", "tome.hip", "vit.hip", "decode.hip", "preprocess.hip"]
COMMON = ["--offload-arch=gfx950", "-O3", "-std=c++17", "-fPIC", "-Wall", "-Wno-unused-function"]
# tome.hip carries the bit-exact fp32 contract: no implicit FMA contraction
# preprocess.hip computes Pillow's resampling taps in doubles on the host: same rule
# attn.hip rescales its output accumulators every key block: keep them in VGPRs (MFMA VGPR form) instead of paying
# 224 v_accvgpr_read/write per block for the AGPR round trip
PER_FILE = {"tome.hip": ["-ffp-contract=off"], "preprocess.hip": ["-ffp-contract=off"],
            "attn.hip": ["-mllvm", "-amdgpu-mfma-vgpr-form=1"]}


def _hipcc() -> str:
    for c in (os.environ.get("HIPCC"), "/opt/rocm/bin/hipcc", "hipcc"):
        if c and (os.path.isabs(c) and os.path.exists(c) or not os.path.isabs(c)):
            return c
    raise RuntimeError("hipcc not found")


def needs_build(out: str = OUT) -> bool:
    """stale when any kernel source, the ABI header or this script (the flags) is newer than `out`"""
    if not os.path.exists(out):
        return True
    t = os.path.getmtime(out)
    deps = [os.path.join(CSRC, f) for f in os.listdir(CSRC)] + [os.path.join(HERE, "..", "include", "aurora_hip.h"), os.path.abspath(__file__)]
    return any(os.path.getmtime(d) > t for d in deps)


# Host-side sanitizer build (ASan/UBSan over the C-ABI host layer): the same sources with AddressSanitizer and
# UndefinedBehaviorSanitizer on the HOST code only (-fno-gpu-sanitize: the gfx950 code objects are the product's), linked against
# the shared ASan runtime so that an un-instrumented python can load it with LD_PRELOAD=<asan_runtime()>.  tests/test_asan_host.py.
OUT_ASAN = os.path.join(HERE, "libaurora_hip_asan.so")
OUT_UBSAN = os.path.join(HERE, "libaurora_hip_ubsan.so")        # UBSan alone: no preload, coexists with the HIP runtime on a GPU box
# Device-side AddressSanitizer: host AND gfx950 code instrumented; needs an xnack+ code object, a GPU
# running with HSA_XNACK=1 and the ASan runtime preloaded.  tools/gpu/device_asan.sh tries it over the cross-workgroup hand-over tests.
OUT_DASAN = os.path.join(HERE, "libaurora_hip_dasan.so")
SAN_DEV = ["-fsanitize=address", "-fgpu-sanitize", "-shared-libsan", "-fno-omit-frame-pointer", "-g"]
SAN = ["-fsanitize=address,undefined", "-fno-gpu-sanitize", "-fno-omit-frame-pointer", "-g", "-shared-libsan"]
SAN_UB = ["-fsanitize=undefined", "-fno-sanitize-recover=undefined", "-fno-gpu-sanitize", "-fno-omit-frame-pointer", "-g", "-shared-libsan"]


def asan_runtime(name: str = "asan") -> str:
    """shared sanitizer runtime to LD_PRELOAD into an un-instrumented python: "asan" or "ubsan_standalone" """
    import glob
    hits = sorted(glob.glob(f"/opt/rocm/lib/llvm/lib/clang/*/lib/linux/libclang_rt.{name}-x86_64.so"))
    if not hits:
        raise RuntimeError(f"libclang_rt.{name}-x86_64.so not found under /opt/rocm/lib/llvm")
    return hits[-1]


def build(force: bool = False, verbose: bool = True, sanitize=False) -> str:
    """sanitize: False (the product), True / "asan" (ASan + UBSan on the host code), "ubsan" (UBSan only, traps made fatal),
    "dasan" (device-side ASan)"""
    ub = sanitize == "ubsan"
    dasan = sanitize == "dasan"
    out = OUT_DASAN if dasan else OUT_UBSAN if ub else (OUT_ASAN if sanitize else OUT)
    san = SAN_DEV if dasan else SAN_UB if ub else SAN
    if not force and not needs_build(out):
        return out
    hipcc = _hipcc()
    objdir = os.path.join(HERE, "build_dasan" if dasan else ("build_ubsan" if ub else "build_asan") if sanitize else "build")
    os.makedirs(objdir, exist_ok=True)
    common = [c if c != "-O3" else "-O1" for c in COMMON] + san if sanitize else COMMON
    arch = "--offload-arch=gfx950:xnack+" if dasan else "--offload-arch=gfx950"
    common = [arch if c == "--offload-arch=gfx950" else c for c in common]

    def compile_one(src):
        obj = os.path.join(objdir, src.replace(".hip", ".o"))
        cmd = [hipcc, *common, *PER_FILE.get(src, []), "-c", os.path.join(CSRC, src), "-o", obj]
        r = subprocess.run(cmd, capture_output=True, text=True)
        if r.returncode != 0:
            raise RuntimeError(f"hipcc failed for {src}:\n{r.stderr[-6000:]}")
        if verbose and r.stderr.strip():
            print(r.stderr[-3000:], file=sys.stderr)
        return obj

    with ThreadPoolExecutor(max_workers=min(8, len(SOURCES))) as ex:
        objs = list(ex.map(compile_one, SOURCES))
    cmd = [hipcc, arch, "-shared", "-fPIC", *(san if sanitize else []), "-o", out, *objs]
    r = subprocess.run(cmd, capture_output=True, text=True)
    if r.returncode != 0:
        raise RuntimeError(f"link failed:\n{r.stderr[-4000:]}")
    if verbose:
        print(f"built {out} ({os.path.getsize(out) / 1e6:.1f} MB)")
    return out


if __name__ == "__main__":
    build(force="--force" in sys.argv, sanitize="dasan" if "--device-asan" in sys.argv else "ubsan" if "--ubsan" in sys.argv else ("--sanitize" in sys.argv))
