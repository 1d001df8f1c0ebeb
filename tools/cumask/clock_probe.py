"""libclock_probe.so (clock_probe.hip: a one-wave kernel that reads the effective shader clock) for the lab scripts of this directory.
(Round 3's contention_lab.py / power_lab.py, which also drove the MFMA forms of the decode attention removed in round 5, are in the
history: `git show 762d157:tools/cumask/contention_lab.py`; their results are profiles/r03_contention_lab.log, r03_power_lab.log.)"""
import ctypes as C
import os
import subprocess


def clock_probe_lib():
    so = os.path.join(os.path.dirname(os.path.abspath(__file__)), "libclock_probe.so")
    if not os.path.exists(so):
        subprocess.check_call(["/opt/rocm/bin/hipcc", "--offload-arch=gfx950", "-O2", "-shared", "-fPIC",
                               os.path.join(os.path.dirname(so), "clock_probe.hip"), "-o", so])
    lib = C.CDLL(so)
    lib.clock_probe_launch.restype = C.c_int
    lib.clock_probe_launch.argtypes = [C.c_void_p, C.c_void_p, C.c_int]
    return lib
