"""Is the mutual slow-down of the two CU-masked streams a POWER effect?  Samples `rocm-smi` (socket power, sclk) while the decode
attention runs alone on the top 16 CUs of every XCD, the prefill gate/up GEMM alone on the bottom 16 (product and no-DMA form), and
both together.  (lab tool)

    python tools/cumask/power_lab.py [B]
"""
import json
import os
import subprocess
import sys
import threading
import time

import torch

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
from aurora_amd import synthetic as S                     # noqa: E402
from aurora_amd.engine import AuroraCapEngine, _rup       # noqa: E402
from aurora_amd.streams import cu_masked_stream           # noqa: E402


def smi():
    try:
        out = subprocess.run(["rocm-smi", "--showpower", "--showclocks", "--json"], capture_output=True, text=True, timeout=20).stdout
        d = json.loads(out)
        card = d[sorted(d)[0]]
        pw = next((v for k, v in card.items() if "ower" in k and "(W)" in k), None)
        sclk = next((v for k, v in card.items() if k.lower().startswith("sclk")), None)
        mclk = next((v for k, v in card.items() if k.lower().startswith("mclk")), None)
        fclk = next((v for k, v in card.items() if k.lower().startswith("fclk")), None)
        return pw, sclk, mclk, fclk
    except Exception as e:                                 # noqa: BLE001
        return ("err", repr(e)[:80], None, None)


def main():
    B = int(sys.argv[1]) if len(sys.argv) > 1 and sys.argv[1].isdigit() else 128
    l = S.VICUNA_7B_16K
    L0 = 2142
    eng = AuroraCapEngine({"vit": None, "llm": l}, {"llm": S.llm_weights(l)}, max_frames=1, max_batch=B, max_ctx=_rup(L0 + 256, 64), max_new_tokens=256)
    torch.cuda.empty_cache()
    eng.begin_batch(B, 256, None)
    g = torch.Generator(device="cuda").manual_seed(0)
    emb0 = (torch.randn(_rup(L0, 32), l["hidden_size"], generator=g, device="cuda") * 0.02).half()
    for b in range(B):
        eng.prefill(b, emb0.clone(), L0)
    torch.cuda.synchronize()
    eng.set_option("microbench_prefill_nseq", 4)
    sd, sf, sa = cu_masked_stream(16, from_top=True), cu_masked_stream(16), torch.cuda.Stream()
    print("idle:", smi(), flush=True)

    def mb(kernel, stream, iters, box, key):
        with torch.cuda.stream(stream):
            box[key] = eng.microbench(kernel, iters)

    def run(jobs, label):
        box = {}
        ths = [threading.Thread(target=mb, args=(k, st, it, box, k + str(i))) for i, (k, st, it) in enumerate(jobs)]
        for t in ths:
            t.start()
        samples = []
        time.sleep(0.5)
        for _ in range(4):
            samples.append(smi())
            time.sleep(0.3)
        for t in ths:
            t.join()
        print(f"{label:60s} us: " + ", ".join(f"{k} {v:.1f}" for k, v in box.items()) + " | smi (W, sclk, mclk, fclk): " + " ".join(str(s) for s in samples), flush=True)

    eng.set_option("gemm_max_wgs", 128)
    if "--stream" in sys.argv:                             # a bare HBM read stream (no MFMA, one add per 16 bytes) beside the decode attention's watts
        import ctypes as C
        from tools.cumask.contention_lab import clock_probe_lib
        lib = clock_probe_lib()
        lib.lab_stream_launch.restype = C.c_int
        lib.lab_stream_launch.argtypes = [C.c_void_p, C.c_void_p, C.c_int64, C.c_int, C.c_int, C.c_int, C.c_void_p]
        big = torch.zeros(8 << 30, dtype=torch.uint8, device="cuda")
        sink = torch.zeros(4, dtype=torch.float32, device="cuda")
        for st, name, blocks in ((sa, "ALL CUs", 2048), (sa, "ALL CUs", 1024), (sd, "top 16 CUs per XCD", 1024)):
            samples = []
            e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            with torch.cuda.stream(st):
                e0.record()
                lib.lab_stream_launch(C.c_void_p(st.cuda_stream), C.c_void_p(big.data_ptr()), 8 << 30, 2500, blocks, 1, C.c_void_p(sink.data_ptr()))
                e1.record()
            time.sleep(0.5)
            for _ in range(4):
                samples.append(smi())
                time.sleep(0.3)
            e1.synchronize()
            print(f"bare nt read stream on {name} ({blocks} wgs): {(8 << 30) * 2500 / (e0.elapsed_time(e1) * 1e-3) / 1e12:.2f} TB/s | smi: " + " ".join(str(x) for x in samples), flush=True)
        run([("dec_attn", sa, 4000)], "decode attention on ALL CUs")
        # the attention's address pattern without its arithmetic, in the pool's layout and with heads interleaved per fragment
        lib.lab_kvpattern_launch.restype = C.c_int
        lib.lab_kvpattern_launch.argtypes = [C.c_void_p, C.c_void_p, C.c_int, C.c_int, C.c_int, C.c_void_p]
        for mode, name in ((0, "[head][fragment] (the KV pool's layout)"), (1, "[fragment][head]")):
            samples = []
            e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            reps, slots, pps = 5000, 128, 36
            with torch.cuda.stream(sa):
                e0.record()
                for _ in range(reps):
                    lib.lab_kvpattern_launch(C.c_void_p(sa.cuda_stream), C.c_void_p(big.data_ptr()), slots, pps, mode, C.c_void_p(sink.data_ptr()))
                e1.record()
            time.sleep(0.5)
            for _ in range(4):
                samples.append(smi())
                time.sleep(0.3)
            e1.synchronize()
            nbytes = reps * slots * pps * (1 << 20)
            print(f"KV address pattern {name}: {nbytes / (e0.elapsed_time(e1) * 1e-3) / 1e12:.2f} TB/s ({e0.elapsed_time(e1) / reps * 1e3:.1f} us per launch) | smi: "
                  + " ".join(str(x) for x in samples), flush=True)
        lib.lab_kvwork_launch.restype = C.c_int
        lib.lab_kvwork_launch.argtypes = [C.c_void_p, C.c_void_p, C.c_int, C.c_int, C.c_int, C.c_void_p]
        for work, name in ((0, "walk only"), (1, "+ 128 v_dot2c per page"), (3, "+ dot2c + softmax exchanges"), (7, "+ dot2c + exchanges + exp + LDS hand-over (= the attention)"),
                           (2, "+ exchanges only"), (4, "+ exp + LDS only")):
            samples = []
            e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            reps, slots, pps = 5000, 128, 36
            with torch.cuda.stream(sa):
                e0.record()
                for _ in range(reps):
                    lib.lab_kvwork_launch(C.c_void_p(sa.cuda_stream), C.c_void_p(big.data_ptr()), slots, pps, work, C.c_void_p(sink.data_ptr()))
                e1.record()
            time.sleep(0.5)
            for _ in range(4):
                samples.append(smi())
                time.sleep(0.3)
            e1.synchronize()
            nbytes = reps * slots * pps * (1 << 20)
            print(f"KV walk {name}: {nbytes / (e0.elapsed_time(e1) * 1e-3) / 1e12:.2f} TB/s ({e0.elapsed_time(e1) / reps * 1e3:.1f} us per launch) | W: "
                  + " ".join(str(x[0]) for x in samples), flush=True)
        eng.close()
        return
    if "--attn-flavors" in sys.argv:                       # energy of the KV stream by load policy / occupancy (same bytes, same kernel body)
        for variant, name in ((1, "MFMA, nt loads, 1 wave/SIMD"), (4, "VALU dot2c, 2 waves/SIMD"), (2, "MFMA, default-policy loads"), (3, "MFMA, nt loads, 2 waves/SIMD"), (0, "MFMA, un-pipelined")):
            eng.set_option("dec_attn_variant", variant)
            run([("dec_attn", sd, 4000)], f"decode attention [{name}] on the top 16 CUs per XCD")
            run([("dec_attn", sa, 4000)], f"decode attention [{name}] on ALL CUs")
        eng.set_option("dec_attn_variant", 1)
        for k in ("dec_qkv", "dec_gateup", "dec_down", "dec_o"):
            run([(k, sa, 40000)], f"{k} on ALL CUs")
        eng.close()
        return
    run([("dec_attn", sd, 5000)], "decode attention alone, top 16 CUs per XCD")
    for lab, name in ((0, "product"), (5, "no operand DMA")):
        eng.set_option("gemm_lab", lab)
        run([("pre_gateup", sf, 2200)], f"prefill gate/up [{name}] alone, bottom 16 CUs per XCD")
        run([("dec_attn", sd, 5500), ("pre_gateup", sf, 1800)], f"both: decode attention + gate/up [{name}]")
    eng.set_option("gemm_lab", 0)
    eng.set_option("gemm_max_wgs", 0)
    run([("pre_gateup", sa, 3000)], "prefill gate/up alone on ALL CUs")
    run([("dec_attn", sa, 5000)], "decode attention alone on ALL CUs")
    eng.close()


if __name__ == "__main__":
    main()
