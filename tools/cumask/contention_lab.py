"""What does the front end's 256x256 GEMM take away from a decode stream running beside it - HBM / fabric bandwidth, or power
(shader clock)?  (lab tool)

Two CU-masked streams as in the serving schedule (decode on the top 16 CUs of every XCD, front end on the bottom 16).  One kernel
loops in a background thread on its stream while the other is timed on its own stream (aur_microbench), then the roles swap; a
one-wave clock probe reads the effective shader clock during the overlap.  The GEMM is the prefill gate/up projection of a 4-clip
pass (M = 8576, N = 22016, K = 4096) in these forms (gemm256.hip G2Lab):
  0 product kernel | 1 A nt | 2 W nt | 3 A+W nt | 4 A+W sc1 | 5 no operand DMA (MFMA + LDS only) | 6 no MFMA (DMA + barriers, burst
  replaced by s_sleep) | 7 A read from a K-tile-major image (32 KiB blocks instead of 128-byte row pieces)
and the decode attention with non-temporal (product) or default-policy KV loads.

    python tools/cumask/contention_lab.py [B] [--quick]
"""
import ctypes as C
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

LAB = {0: "product", 1: "A nt", 2: "W nt", 3: "A+W nt", 4: "A+W sc1", 5: "no DMA", 6: "no MFMA", 7: "A tiled"}


def clock_probe_lib():
    so = os.path.join(os.path.dirname(os.path.abspath(__file__)), "libclock_probe.so")
    if not os.path.exists(so):
        subprocess.check_call(["/opt/rocm/bin/hipcc", "--offload-arch=gfx950", "-O2", "-shared", "-fPIC",
                               os.path.join(os.path.dirname(so), "clock_probe.hip"), "-o", so])
    lib = C.CDLL(so)
    lib.clock_probe_launch.restype = C.c_int
    lib.clock_probe_launch.argtypes = [C.c_void_p, C.c_void_p, C.c_int]
    return lib


def main():
    B = int(sys.argv[1]) if len(sys.argv) > 1 and sys.argv[1].isdigit() else 128
    quick = "--quick" in sys.argv
    l = S.VICUNA_7B_16K
    L0 = 2142
    eng = AuroraCapEngine({"vit": None, "llm": l}, {"llm": S.llm_weights(l)}, max_frames=1, max_batch=B, max_ctx=_rup(L0 + 256, 64),
                          max_new_tokens=256)
    torch.cuda.empty_cache()
    d = l["hidden_size"]
    eng.begin_batch(B, 256, None)
    g = torch.Generator(device="cuda").manual_seed(0)
    emb0 = (torch.randn(_rup(L0, 32), d, generator=g, device="cuda") * 0.02).half()
    for b in range(B):
        eng.prefill(b, emb0.clone(), L0)
    torch.cuda.synchronize()
    eng.set_option("microbench_prefill_nseq", 4)
    sd = cu_masked_stream(16, from_top=True)
    sf = cu_masked_stream(16)
    free = torch.cuda.Stream()
    eng.set_option("gemm_max_wgs", 128)
    probe = clock_probe_lib()
    pout = torch.zeros(2, dtype=torch.int64, device="cuda")

    def mb(kernel, stream, iters):
        with torch.cuda.stream(stream):
            return eng.microbench(kernel, iters)

    def mhz(spin_us=3000):
        pout.zero_()
        torch.cuda.current_stream().synchronize()
        rc = probe.clock_probe_launch(C.c_void_p(free.cuda_stream), C.c_void_p(pout.data_ptr()), spin_us)
        assert rc == 0, rc
        free.synchronize()
        c, w = pout.tolist()
        return 100.0 * c / max(w, 1)

    def pair(xk, xs, yk, ys, x_alone, y_alone, span=1.6):
        """x timed beside a looping y, then y timed beside a looping x; the clock during the first overlap"""
        res = {}
        for (fk, fs, fa, bk, bs, ba, tag) in ((xk, xs, x_alone, yk, ys, y_alone, "x"), (yk, ys, y_alone, xk, xs, x_alone, "y")):
            box = {}
            th = threading.Thread(target=lambda: box.setdefault("bg", mb(bk, bs, max(8, int(span * 1e6 / ba)))))
            th.start()
            time.sleep(0.25)
            if tag == "x":
                res["mhz"] = mhz()
            res[tag] = mb(fk, fs, max(8, int(0.45 * span * 1e6 / fa)))
            th.join()
        return res

    print(f"# B = {B} slots at context {L0}; decode stream = top 16 CUs per XCD, front-end stream = bottom 16; GEMM = prefill gate/up, M = {4 * _rup(L0, 32)}", flush=True)
    print(f"idle clock {mhz():.0f} MHz", flush=True)
    alone = {}
    for k, st, it in (("dec_attn", sd, 400), ("dec_gateup", sd, 1500), ("dec_qkv", sd, 1500), ("dec_down", sd, 1500), ("pre_attn", sf, 200)):
        alone[k] = mb(k, st, it)
        print(f"alone {k:12s} {alone[k]:9.1f} us", flush=True)
    eng.set_option("dec_attn_variant", 2)
    alone["dec_attn_plain"] = mb("dec_attn", sd, 400)
    eng.set_option("dec_attn_variant", 1)
    print(f"alone dec_attn (default-policy KV loads) {alone['dec_attn_plain']:9.1f} us", flush=True)
    galone = {}
    for lab in LAB:
        eng.set_option("gemm_lab", lab)
        galone[lab] = mb("pre_gateup", sf, 120)
        box = {}
        th = threading.Thread(target=lambda: box.setdefault("bg", mb("pre_gateup", sf, 400)))
        th.start()
        time.sleep(0.2)
        f = mhz()
        th.join()
        print(f"alone pre_gateup [{lab} {LAB[lab]:8s}] {galone[lab]:9.1f} us   clock while it runs {f:.0f} MHz", flush=True)
    box = {}
    th = threading.Thread(target=lambda: box.setdefault("bg", mb("dec_attn", sd, 1200)))
    th.start()
    time.sleep(0.2)
    print(f"clock while dec_attn runs alone {mhz():.0f} MHz", flush=True)
    th.join()

    print("# together: decode kernel on the decode stream, GEMM variant on the front-end stream (us, and ratio to alone)", flush=True)
    dks = ("dec_attn",) if quick else ("dec_attn", "dec_gateup", "dec_qkv", "dec_down")
    for dk in dks:
        for lab in (LAB if not quick else ()):
            eng.set_option("gemm_lab", lab)
            r = pair(dk, sd, "pre_gateup", sf, alone[dk], galone[lab])
            print(f"{dk:11s} {r['x']:8.1f} us ({r['x'] / alone[dk]:4.2f}x) | pre_gateup [{lab} {LAB[lab]:8s}] {r['y']:8.1f} us ({r['y'] / galone[lab]:4.2f}x) | "
                  f"clock {r['mhz']:.0f} MHz", flush=True)
    # default-policy KV loads against the product GEMM and the nt GEMM
    eng.set_option("dec_attn_variant", 2)
    for lab in ((0, 3) if not quick else ()):
        eng.set_option("gemm_lab", lab)
        r = pair("dec_attn", sd, "pre_gateup", sf, alone["dec_attn_plain"], galone[lab])
        print(f"dec_attn(plain KV loads) {r['x']:8.1f} us ({r['x'] / alone['dec_attn_plain']:4.2f}x) | pre_gateup [{lab} {LAB[lab]:8s}] {r['y']:8.1f} us "
              f"({r['y'] / galone[lab]:4.2f}x) | clock {r['mhz']:.0f} MHz", flush=True)
    eng.set_option("dec_attn_variant", 1)
    eng.set_option("gemm_lab", 0)
    # the other front-end kernels beside the decode attention
    for fk in (("pre_attn", "pre_qkv", "pre_down", "pre_o") if not quick else ()):
        if fk not in alone:
            alone[fk] = mb(fk, sf, 100)
        r = pair("dec_attn", sd, fk, sf, alone["dec_attn"], alone[fk])
        print(f"dec_attn    {r['x']:8.1f} us ({r['x'] / alone['dec_attn']:4.2f}x) | {fk:10s} {r['y']:8.1f} us ({r['y'] / alone[fk]:4.2f}x, alone {alone[fk]:.1f}) | clock {r['mhz']:.0f} MHz",
              flush=True)
    # decode attention variants: 1 (one wave per SIMD, K double-buffered in registers) vs 3 (two waves per SIMD)
    eng.set_option("gemm_tile_order", 1)
    for variant in (1, 3, 4):
        eng.set_option("dec_attn_variant", variant)
        eng.set_option("gemm_max_wgs", 128)
        a16 = mb("dec_attn", sd, 400)
        a_all = mb("dec_attn", free, 400)
        a20 = mb("dec_attn", cu_masked_stream(20, from_top=True), 400)
        a12 = mb("dec_attn", cu_masked_stream(12, from_top=True), 400)
        r = pair("dec_attn", sd, "pre_gateup", sf, a16, galone[0])
        r2 = pair("dec_attn", sd, "pre_attn", sf, a16, alone["pre_attn"])
        print(f"dec_attn variant {variant}: alone all CUs {a_all:.1f} us, top 20 / 16 / 12 CUs per XCD {a20:.1f} / {a16:.1f} / {a12:.1f} us | beside gate/up {r['x']:.1f} us "
              f"(GEMM {r['y']:.1f}) | beside prefill attention {r2['x']:.1f} us (attn {r2['y']:.1f})", flush=True)
    eng.set_option("dec_attn_variant", 1)
    if "--attn-only" in sys.argv:
        eng.close()
        return
    # tile order of the product GEMM beside the decode attention (0 per-XCD ranges, 1 compact shared blocks, 2 hand-down)
    for fk in ("pre_gateup", "pre_qkv", "pre_down", "pre_o"):
        for order in (0, 1, 2):
            eng.set_option("gemm_tile_order", order)
            ga = mb(fk, sf, 120)
            r = pair("dec_attn", sd, fk, sf, alone["dec_attn"], ga)
            print(f"tile_order {order}: dec_attn {r['x']:8.1f} us ({r['x'] / alone['dec_attn']:4.2f}x) | {fk:10s} {r['y']:8.1f} us (alone {ga:.1f})", flush=True)
    eng.set_option("gemm_tile_order", 1)
    eng.close()


if __name__ == "__main__":
    main()
