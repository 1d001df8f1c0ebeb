"""The persistent 256x256 GEMM's tile order (aurora_amd/csrc/tile_order.h) is plain C++: compile its host check with g++ and
run it - every block id must map onto every tile exactly once, for whole blocks, strips and grids of any size."""
import os
import subprocess

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def test_tile_order_is_a_bijection(tmp_path):
    exe = tmp_path / "tile_order_check"
    subprocess.run(["g++", "-O2", "-std=c++17", "-I", os.path.join(ROOT, "aurora_amd", "csrc"),
                    os.path.join(ROOT, "tests", "native", "tile_order_check.cpp"), "-o", str(exe)], check=True, timeout=300)
    r = subprocess.run([str(exe)], capture_output=True, text=True, timeout=300)
    assert r.returncode == 0 and r.stdout.startswith("ok"), r.stdout + r.stderr
