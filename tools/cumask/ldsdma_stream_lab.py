"""Does a CU keep more bytes in flight when they land in LDS instead of registers?  (lab tool, round 6)

The decode attention on 16 CUs of every XCD (the masked steps of the serving loop) streams 5.97 TB/s = 47 GB/s per CU whatever its
pipeline; a bare register-load walk of the same addresses reaches about the same.  This lab walks the same KV addresses with
`global_load_lds` (nothing consumed) with 1 / 2 / 4 waves per workgroup, on the whole chip and on 16 CUs per XCD, beside the bare
register-load walk (lab_kvpattern_kernel).

    python tools/cumask/ldsdma_stream_lab.py
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
    lib.lab_kvpattern_launch.restype = C.c_int
    lib.lab_kvpattern_launch.argtypes = [C.c_void_p, C.c_void_p, C.c_int, C.c_int, C.c_int, C.c_void_p]
    lib.lab_kvpattern_lds_launch.restype = C.c_int
    lib.lab_kvpattern_lds_launch.argtypes = [C.c_void_p, C.c_void_p, C.c_int, C.c_int, C.c_int, C.c_void_p]
    slots, pages = 128, 34                                        # 128 x 34 pages of 1 MiB: the pool of a 2142-token context, one layer
    pool = torch.zeros(slots * pages << 20, dtype=torch.uint8, device="cuda")
    pool.view(torch.int16).fill_(0x3c00)
    sink = torch.zeros(4, dtype=torch.float32, device="cuda")
    nbytes = slots * pages * 32 * 32768                           # every (slot, head) reads 32 KiB per page
    for cus in (0, 16):
        st = cu_masked_stream(cus, from_top=True) if cus else torch.cuda.Stream()
        def run(fn, reps=20):
            e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            with torch.cuda.stream(st):
                for _ in range(3):
                    assert fn() == 0
                e0.record()
                for _ in range(reps):
                    assert fn() == 0
                e1.record()
            e1.synchronize()
            us = e0.elapsed_time(e1) * 1e3 / reps
            return us, nbytes / us / 1e6
        sp, pp, kp = C.c_void_p(st.cuda_stream), C.c_void_p(pool.data_ptr()), C.c_void_p(sink.data_ptr())
        for rep in range(2):
            us, tbs = run(lambda: lib.lab_kvpattern_launch(sp, pp, slots, pages, 0, kp))
            print(f"cus/xcd={cus or 32} register loads (32 in flight per wave)      {us:8.1f} us  {tbs:5.2f} TB/s", flush=True)
            for waves in (1, 2, 4):
                us, tbs = run(lambda: lib.lab_kvpattern_lds_launch(sp, pp, slots, pages, waves, kp))
                print(f"cus/xcd={cus or 32} LDS-DMA, {waves} wave(s) per workgroup, 32 KiB ring each {us:8.1f} us  {tbs:5.2f} TB/s", flush=True)


if __name__ == "__main__":
    main()
