"""Does the 256 MiB memory-side cache turn the 8 XCDs' reads of the SAME bytes into one HBM read - when they ask at the same moment,
and when they ask one after another?  (lab tool; decides whether the GEMM's tile order should hand operand tiles down from XCD to
XCD round by round instead of sharing them within a round.)

Every XCD sweeps the whole buffer once (8 x the bytes cross the fabric, 1 x is unique).  Reported: aggregate TB/s over the fabric.
An HBM-bound sweep cannot exceed ~5.5-7 TB/s; anything above it is served by the memory-side cache.

    python tools/cumask/share_lab.py
"""
import ctypes as C
import os
import sys

import torch

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
from aurora_amd.streams import cu_masked_stream           # noqa: E402
from tools.cumask.clock_probe import clock_probe_lib   # noqa: E402


def main():
    lib = clock_probe_lib()
    lib.lab_share_launch.restype = C.c_int
    lib.lab_share_launch.argtypes = [C.c_void_p, C.c_void_p, C.c_int64, C.c_int64, C.c_int, C.c_void_p]
    lib.lab_stream_launch.restype = C.c_int
    lib.lab_stream_launch.argtypes = [C.c_void_p, C.c_void_p, C.c_int64, C.c_int, C.c_int, C.c_int, C.c_void_p]
    big = torch.zeros(8 << 30, dtype=torch.uint8, device="cuda")
    flush = torch.zeros(1 << 30, dtype=torch.uint8, device="cuda")
    sink = torch.zeros(4, dtype=torch.float32, device="cuda")
    streams = {"all CUs": torch.cuda.Stream(), "16 CUs per XCD": cu_masked_stream(16)}

    def run(st, nbytes, stagger, blocks):
        flush.add_(1)                                               # 1 GiB of other traffic: nothing of `big` is cached when a run starts
        torch.cuda.synchronize()
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        with torch.cuda.stream(st):
            e0.record()
            rc = lib.lab_share_launch(C.c_void_p(st.cuda_stream), C.c_void_p(big.data_ptr()), nbytes, stagger, blocks, C.c_void_p(sink.data_ptr()))
            assert rc == 0
            e1.record()
        e1.synchronize()
        return 8 * nbytes / (e0.elapsed_time(e1) * 1e-3) / 1e12

    for sname, st in streams.items():
        print(f"# {sname}", flush=True)
        for blocks in (64, 128, 512, 2048):                        # 8 / 16 / 64 / 256 workgroups per XCD: rate control
            row = []
            for stagger_mib in (0, 1, 16, 64, 128):
                row.append(f"stagger {stagger_mib:3d} MiB: {run(st, 4 << 30, stagger_mib << 20, blocks):5.2f}")
            print(f"{blocks:5d} wgs | " + " | ".join(row) + " TB/s (fabric)", flush=True)


if __name__ == "__main__":
    main()
