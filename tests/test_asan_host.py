"""Sanitizer pass over the C-ABI host layer: libaurora_hip rebuilt with AddressSanitizer +
UndefinedBehaviorSanitizer on its host code (aurora_amd/build.py --sanitize; device code untouched) and walked from an
un-instrumented python with the ASan runtime preloaded.  CPU: every entry point that needs no GPU, including the error paths.
GPU (-m gpu): a tiny engine through ViT + ToMe + prefill + decode, batch mode and both continuous-batching schedules."""
import os
import subprocess
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _run(args, timeout):
    from aurora_amd import build as B
    so = B.build(sanitize=True, verbose=False)
    env = dict(os.environ, LD_PRELOAD=B.asan_runtime(), AURORA_HIP_SO=so, PYTHONPATH=ROOT,
               # python itself leaks by design; the HIP runtime maps memory inside ASan's shadow gap
               ASAN_OPTIONS="detect_leaks=0:protect_shadow_gap=0:halt_on_error=1:abort_on_error=0",
               UBSAN_OPTIONS="print_stacktrace=1:halt_on_error=1")
    r = subprocess.run([sys.executable, os.path.join(ROOT, "tests", "asan_host_script.py")] + args, cwd=ROOT, env=env, capture_output=True,
                       text=True, timeout=timeout)
    report = r.stdout[-3000:] + r.stderr[-6000:]
    assert "AddressSanitizer" not in r.stderr and "runtime error:" not in r.stderr, report
    assert r.returncode == 0, report
    return r.stdout


def test_host_only_entry_points_under_asan_ubsan():
    assert "host-only ABI walk ok" in _run([], 900)


@pytest.mark.gpu
def test_engine_walk_under_asan_ubsan():
    assert "engine walk under the host sanitizers ok" in _run(["gpu"], 1500)
