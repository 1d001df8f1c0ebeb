"""Is the resource the decode stream and the front end fight over HBM, or the fabric in front of it?  (lab tool)

The decode attention (HBM stream, top 16 CUs of every XCD) runs beside a plain read streamer on the bottom 16 CUs whose buffer
is either a few tens of MiB - larger than an XCD's share of L2, smaller than the 256 MiB memory-side cache, so that after the
first sweep its reads are Infinity-Cache hits - or several GiB (HBM).  If cache-resident traffic of the same rate slows the
decode as much as HBM traffic does, the shared bottleneck is the fabric; if it does not, operand reuse through the memory-side
cache is worth engineering for.

    python tools/cumask/mall_lab.py [B]
"""
import ctypes as C
import os
import sys
import threading
import time

import torch

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
from aurora_amd import synthetic as S                     # noqa: E402
from aurora_amd.engine import AuroraCapEngine, _rup       # noqa: E402
from aurora_amd.streams import cu_masked_stream           # noqa: E402
from tools.cumask.clock_probe import clock_probe_lib   # noqa: E402


def main():
    B = int(sys.argv[1]) if len(sys.argv) > 1 and sys.argv[1].isdigit() else 128
    l = S.VICUNA_7B_16K
    L0 = 2142
    eng = AuroraCapEngine({"vit": None, "llm": l}, {"llm": S.llm_weights(l)}, max_frames=1, max_batch=B, max_ctx=_rup(L0 + 256, 64),
                          max_new_tokens=256)
    torch.cuda.empty_cache()
    eng.begin_batch(B, 256, None)
    g = torch.Generator(device="cuda").manual_seed(0)
    emb0 = (torch.randn(_rup(L0, 32), l["hidden_size"], generator=g, device="cuda") * 0.02).half()
    for b in range(B):
        eng.prefill(b, emb0.clone(), L0)
    torch.cuda.synchronize()
    sd = cu_masked_stream(16, from_top=True)
    sf = cu_masked_stream(16)
    lib = clock_probe_lib()
    lib.lab_stream_launch.restype = C.c_int
    lib.lab_stream_launch.argtypes = [C.c_void_p, C.c_void_p, C.c_int64, C.c_int, C.c_int, C.c_int, C.c_void_p]
    big = torch.zeros(6 << 30, dtype=torch.uint8, device="cuda")
    sink = torch.zeros(4, dtype=torch.float32, device="cuda")
    kv_bytes = B * (L0 + 1) * 2 * l["hidden_size"] * 2

    def mb(kernel, stream, iters):
        with torch.cuda.stream(stream):
            return eng.microbench(kernel, iters)

    def stream_run(nbytes, reps, blocks, nt):
        """one launch on the front-end stream; returns achieved TB/s"""
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        with torch.cuda.stream(sf):
            e0.record()
            rc = lib.lab_stream_launch(C.c_void_p(sf.cuda_stream), C.c_void_p(big.data_ptr()), nbytes, reps, blocks, nt, C.c_void_p(sink.data_ptr()))
            assert rc == 0, rc
            e1.record()
        e1.synchronize()
        return nbytes * reps / (e0.elapsed_time(e1) * 1e-3) / 1e12

    a_attn = mb("dec_attn", sd, 400)
    print(f"# B = {B}, context {L0}: dec_attn alone on the top 16 CUs per XCD {a_attn:.1f} us = {kv_bytes / a_attn / 1e6:.2f} TB/s", flush=True)
    print("# streamer on the bottom 16 CUs per XCD: buffer MiB, nt, workgroups | alone TB/s | beside dec_attn TB/s | dec_attn us beside it (TB/s) | sum TB/s", flush=True)
    for mib in (64, 128, 192, 4096):
        for nt in (0, 1):
            for blocks in (128, 512, 1024):
                nbytes = mib << 20
                target = 1.5e12 if blocks == 128 else 5e12                      # bytes for ~1 s of streaming
                reps = max(2, int(target / nbytes))
                stream_run(nbytes, 2, blocks, nt)                               # warm the cache
                alone = stream_run(nbytes, max(2, reps // 4), blocks, nt)
                box = {}
                th = threading.Thread(target=lambda: box.setdefault("tbs", stream_run(nbytes, int(reps * 2.0), blocks, nt)))
                th.start()
                time.sleep(0.15)
                t_attn = mb("dec_attn", sd, 1000)
                th.join()
                d_tbs = kv_bytes / t_attn / 1e6
                print(f"{mib:5d} MiB nt={nt} wgs={blocks:4d} | {alone:5.2f} | {box['tbs']:5.2f} | {t_attn:7.1f} us ({d_tbs:4.2f}) | {box['tbs'] + d_tbs:5.2f}", flush=True)
    # ---- CU split: how do the two sides scale with their share of every XCD?
    eng.set_option("microbench_prefill_nseq", 4)
    print("# CU split (CUs per XCD): decode kernels alone on the top k, prefill gate/up alone on the bottom 32 - k, then together", flush=True)
    for k in (8, 12, 16, 20):
        s_d = cu_masked_stream(k, from_top=True)
        s_f = cu_masked_stream(32 - k)
        eng.set_option("gemm_max_wgs", 8 * (32 - k))
        da = {kn: mb(kn, s_d, it) for kn, it in (("dec_attn", 300), ("dec_gateup", 1000), ("dec_qkv", 1000), ("dec_down", 1000), ("dec_o", 1000))}
        ga = mb("pre_gateup", s_f, 100)
        box = {}
        th = threading.Thread(target=lambda: box.setdefault("g", mb("pre_gateup", s_f, max(8, int(2.2e6 / ga)))))
        th.start()
        time.sleep(0.2)
        dt = mb("dec_attn", s_d, max(8, int(0.8e6 / da["dec_attn"])))
        th.join()
        th = threading.Thread(target=lambda: box.setdefault("d", mb("dec_attn", s_d, max(8, int(2.2e6 / da["dec_attn"])))))
        th.start()
        time.sleep(0.2)
        gt = mb("pre_gateup", s_f, max(8, int(0.8e6 / ga)))
        th.join()
        print(f"decode on {k:2d} / front on {32 - k:2d}: alone " + " ".join(f"{kn[4:]} {v:.1f}" for kn, v in da.items()) + f" | pre_gateup {ga:.1f} || together dec_attn {dt:.1f} "
              f"({dt / da['dec_attn']:.2f}x) pre_gateup {gt:.1f} ({gt / ga:.2f}x)", flush=True)
    eng.close()


if __name__ == "__main__":
    main()
