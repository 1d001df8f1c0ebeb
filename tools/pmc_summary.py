#!/usr/bin/env python3
"""Per-kernel, per-launch-shape summary of the three rocprofv3 --pmc passes written by tools/profile_round.sh.

    python tools/pmc_summary.py <prof_dir> <out.json>

For every (kernel, grid size) of our kernels:
  hbm_bytes_per_launch = (2 * FETCH_SIZE + WRITE_SIZE) * 1024      (KiB counters; gfx950 counts wide coalesced reads at half
                                                                    their bytes - MI355X_MICROARCH.md, HBM section)
  mfma_util            = SQ_VALU_MFMA_BUSY_CYCLES / ((GRBM_GUI_ACTIVE / 8 XCDs) * 256 CUs * 4 SIMDs)
  clock_ghz            = (GRBM_GUI_ACTIVE / 8 XCDs) / kernel duration of the same dispatch (effective shader clock)
Calibration of the two units on this rocprofv3 (profiles/r02a): GRBM_GUI_ACTIVE is the SUM over the 8 XCDs (an HBM-bound kernel
reads 19.4 "GHz" = 8 x 2.43); SQ_VALU_MFMA_BUSY_CYCLES advances 16 per v_mfma_f32_16x16x32_f16 summed over all SIMDs (the
5762-tile gate/up GEMM issues 1.888e8 MFMAs and reads 3.01e9), i.e. 100 % = every one of the 1024 SIMDs issuing back to back.
Counter files differ a little between rocprofv3 builds; columns are looked up by the names that exist."""
import csv
import glob
import json
import os
import sys
from collections import defaultdict

CUS = 256
XCDS = 8


def col(r, *names):
    for n in names:
        if n in r and r[n] not in (None, ""):
            return r[n]
    return None


def read_pass(d):
    """-> {(kernel, grid): {counter: [sum, n], "_dur_ns": [sum, n]}}; durations joined from the kernel trace by dispatch id."""
    dur = {}
    for f in glob.glob(os.path.join(d, "**", "*kernel_trace.csv"), recursive=True):
        for r in csv.DictReader(open(f)):
            did = col(r, "Dispatch_Id", "Dispatch_ID")
            if did is not None:
                dur[did] = float(r["End_Timestamp"]) - float(r["Start_Timestamp"])
    acc = defaultdict(lambda: defaultdict(lambda: [0.0, 0]))
    for f in glob.glob(os.path.join(d, "**", "*counter_collection.csv"), recursive=True):
        seen = set()
        for r in csv.DictReader(open(f)):
            k = col(r, "Kernel_Name", "Kernel Name")
            g = col(r, "Grid_Size", "Grid_Size_X") or "?"
            key = (k, str(g))
            a = acc[key][r["Counter_Name"]]
            a[0] += float(r["Counter_Value"])
            a[1] += 1
            did = col(r, "Dispatch_Id", "Dispatch_ID")
            if did is not None and (did, r["Counter_Name"]) not in seen and r["Counter_Name"] in ("GRBM_GUI_ACTIVE", "FETCH_SIZE", "WRITE_SIZE"):
                seen.add((did, r["Counter_Name"]))
                t = None
                if col(r, "Start_Timestamp") and col(r, "End_Timestamp"):
                    t = float(r["End_Timestamp"]) - float(r["Start_Timestamp"])
                elif did in dur:
                    t = dur[did]
                if t is not None:
                    b = acc[key]["_dur_ns:" + r["Counter_Name"]]
                    b[0] += t
                    b[1] += 1
    return acc


def build_id():
    """sha256[:16] of the library and of bench.py the passes ran (the snapshot on the GPU box has no .git): bench.py compares it with
    the library IT loaded, so that `roofline.traffic` says whether the counters come from the same build"""
    import hashlib
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    out = {}
    for name, rel in (("lib_sha16", os.path.join("aurora_amd", "libaurora_hip.so")), ("bench_sha16", "bench.py")):
        try:
            out[name] = hashlib.sha256(open(os.path.join(root, rel), "rb").read()).hexdigest()[:16]
        except OSError:
            out[name] = None
    return out


def main():
    root, out = sys.argv[1], sys.argv[2]
    batch_x_ctx = float(sys.argv[3]) if len(sys.argv) > 3 else 64 * 2145.0     # decode attention: sequences x mean cached tokens of the pass
    F, W, M = (read_pass(os.path.join(root, p)) for p in ("pmc_fetch", "pmc_write", "pmc_mfma"))
    # optional fourth pass (round 6): TCC_EA0_RDREQ_LEVEL_sum / TCC_EA0_RDREQ_sum = mean L2-to-fabric read latency in L2 clocks - the only
    # handle this rocprofv3 gives on WHERE a kernel's fabric reads are served (no DRAM-side counter: profiles/r06_rocprof_ea_counters.txt):
    # a stream from HBM (the decode attention) is the yardstick, reads served by the memory-side cache come back sooner
    E = read_pass(os.path.join(root, "pmc_ealat")) if os.path.isdir(os.path.join(root, "pmc_ealat")) else {}
    ours = lambda k: k and not (k.startswith("void at::") or "rocprim" in k or k.startswith("__amd_rocclr") or "hipcub" in k)
    keys = sorted({k for k in list(F) + list(W) + list(M) + list(E) if ours(k[0])})
    res = {"_note": "per (kernel, grid): means over the dispatches of `bench.py --batch B --prefill-group 4 --batch-mode --steps 1 --warmup 0 --max_new_tokens 6 --no-graph` "
                    "(B x 2145 = _batch_x_ctx: the decode attention's sequences x cached tokens in this pass) under three "
                    "separate rocprofv3 --pmc passes; hbm_bytes = (2*FETCH_SIZE + WRITE_SIZE)*1024 (gfx950 FETCH half-count correction); "
                    "mfma_util = SQ_VALU_MFMA_BUSY_CYCLES / (GRBM_GUI_ACTIVE / 8 XCDs * 256 CUs * 4 SIMDs); clock_ghz = GRBM_GUI_ACTIVE / 8 / duration",
           "_batch_x_ctx": batch_x_ctx, "_build": build_id(), "kernels": []}
    mean = lambda a, c: (a[c][0] / a[c][1]) if c in a and a[c][1] else None
    for k in keys:
        f, w, m = F.get(k, {}), W.get(k, {}), M.get(k, {})
        ea = E.get(k, {})
        lvl, req = mean(ea, "TCC_EA0_RDREQ_LEVEL_sum"), mean(ea, "TCC_EA0_RDREQ_sum")
        e = {"kernel": k[0], "grid": k[1], "dispatches": int(max([v[1] for a in (f, w, m) for c, v in a.items() if not c.startswith("_")] or [0]))}
        fk, wk = mean(f, "FETCH_SIZE"), mean(w, "WRITE_SIZE")
        if fk is not None:
            e["fetch_kib"] = fk
        if wk is not None:
            e["write_kib"] = wk
        if fk is not None and wk is not None:
            e["hbm_bytes_per_launch"] = (2 * fk + wk) * 1024
        busy, gui = mean(m, "SQ_VALU_MFMA_BUSY_CYCLES"), mean(m, "GRBM_GUI_ACTIVE")
        if busy is not None and gui:
            e["mfma_busy_cycles"], e["gui_active_cycles"] = busy, gui
            e["mfma_util"] = busy / (gui / XCDS * CUS * 4)
            d = mean(m, "_dur_ns:GRBM_GUI_ACTIVE")
            if d:
                e["avg_us_under_pmc"] = d / 1e3
                e["clock_ghz"] = gui / XCDS / d
        if lvl is not None and req:
            e["ea_read_latency_clk"] = lvl / req
            e["ea_read_requests"] = req
        res["kernels"].append(e)
    res["kernels"].sort(key=lambda e: -(e.get("avg_us_under_pmc", 0) * e["dispatches"]))
    json.dump(res, open(out, "w"), indent=1)
    print(f"{'kernel':70s} {'grid':>10s} {'n':>6s} {'us':>9s} {'HBM MB':>9s} {'mfma%':>6s} {'GHz':>5s} {'EA rd lat':>9s}")
    for e in res["kernels"][:40]:
        print(f"{e['kernel'][:70]:70s} {e['grid']:>10s} {e['dispatches']:6d} {e.get('avg_us_under_pmc', 0):9.1f} "
              f"{e.get('hbm_bytes_per_launch', 0) / 1e6:9.1f} {100 * e.get('mfma_util', 0):6.1f} {e.get('clock_ghz', 0):5.2f} {e.get('ea_read_latency_clk', 0):9.0f}")


if __name__ == "__main__":
    main()
