"""Sanitizer pass over the C-ABI host layer: libaurora_hip rebuilt with AddressSanitizer +
UndefinedBehaviorSanitizer on its host code (aurora_amd/build.py --sanitize; device code untouched) and walked from an
un-instrumented python with the ASan runtime preloaded.  CPU: every entry point that needs no GPU, including the error paths.
GPU (-m gpu): a tiny engine through ViT + ToMe + prefill + decode, batch mode and both continuous-batching schedules."""
import os
import subprocess
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _run(args, timeout, mode="asan", check=True):
    from aurora_amd import build as B
    so = B.build(sanitize=mode, verbose=False)
    env = dict(os.environ, AURORA_HIP_SO=so, PYTHONPATH=ROOT, UBSAN_OPTIONS="print_stacktrace=1:halt_on_error=1")
    env["LD_PRELOAD"] = B.asan_runtime("asan" if mode == "asan" else "ubsan_standalone")
    if mode == "asan":
        # python itself leaks by design; the HIP runtime maps memory inside ASan's shadow gap
        env["ASAN_OPTIONS"] = "detect_leaks=0:protect_shadow_gap=0:halt_on_error=1:abort_on_error=0"
    r = subprocess.run([sys.executable, os.path.join(ROOT, "tests", "asan_host_script.py")] + args, cwd=ROOT, env=env, capture_output=True,
                       text=True, timeout=timeout)
    report = r.stdout[-3000:] + r.stderr[-6000:]
    if check:
        assert "AddressSanitizer" not in r.stderr and "runtime error:" not in r.stderr, report
        assert r.returncode == 0, report
    return r


def test_host_only_entry_points_under_asan_ubsan():
    assert "host-only ABI walk ok" in _run([], 900).stdout
    assert "host-only ABI walk ok" in _run([], 900, mode="ubsan").stdout


@pytest.mark.gpu
def test_engine_walk_under_the_host_sanitizers():
    """UBSan (standalone runtime preloaded, traps fatal) always; ASan too where its runtime can live beside the HIP runtime:
    ROCm's ASan runtime intercepts hsa_amd_memory_pool_allocate for DEVICE-side ASan and aborts inside that interceptor on a box
    without xnack (measured on the MI355X test box: "out of memory" in asan_interceptors.cpp:804 before any of our code runs) -
    that outcome is reported and tolerated, any finding in aurora code is not."""
    assert "engine walk under the host sanitizers ok" in _run(["gpu"], 1500, mode="ubsan").stdout
    r = _run(["gpu"], 1500, mode="asan", check=False)
    report = r.stdout[-2000:] + r.stderr[-6000:]
    if r.returncode != 0 and "hsa_amd_memory_pool_allocate" in r.stderr and "libaurora_hip" not in r.stderr.split("SUMMARY")[0].split("#1")[0]:
        print("ASan runtime cannot coexist with the HIP runtime on this box (aborts in its own HSA interceptor); UBSan walk passed")
        return
    assert "AddressSanitizer" not in r.stderr and "runtime error:" not in r.stderr and r.returncode == 0, report
