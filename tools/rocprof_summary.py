#!/usr/bin/env python3
"""Summarise rocprofv3 output directories into small text files kept under profiles/.

  kernel stats : <dir>/**/**kernel_stats.csv  -> top kernels by total time (calls, avg us, %)
  kernel shapes: <dir>/**/**kernel_trace.csv  -> the same per (kernel, grid size)
  pmc counters : <dir>/**/**counter_collection.csv -> per-kernel mean of each counter
"""
import csv
import glob
import os
import sys
from collections import defaultdict


def short(name, n=110):
    return name if len(name) <= n else name[: n - 3] + "..."


def kernel_stats(d, out):
    files = glob.glob(os.path.join(d, "**", "*kernel_stats.csv"), recursive=True)
    if not files:
        out.write("no kernel_stats.csv found\n")
        return
    rows = list(csv.DictReader(open(files[0])))
    tot = sum(float(r["TotalDurationNs"]) for r in rows)
    out.write(f"# {files[0]}\n# total kernel time {tot / 1e6:.2f} ms over {sum(int(r['Calls']) for r in rows)} launches\n")
    out.write(f"{'kernel':112s} {'calls':>8s} {'avg_us':>10s} {'total_ms':>10s} {'pct':>6s}\n")
    for r in sorted(rows, key=lambda r: -float(r["TotalDurationNs"]))[:30]:
        out.write(f"{short(r['Name']):112s} {int(r['Calls']):8d} {float(r['AverageNs']) / 1e3:10.2f} "
                  f"{float(r['TotalDurationNs']) / 1e6:10.2f} {100 * float(r['TotalDurationNs']) / tot:6.2f}\n")


def kernel_shapes(d, out):
    """Per-launch trace grouped by (kernel, grid): separates the GEMM shapes that share one kernel name."""
    files = glob.glob(os.path.join(d, "**", "*kernel_trace.csv"), recursive=True)
    if not files:
        out.write("no kernel_trace.csv found\n")
        return
    acc = defaultdict(lambda: [0.0, 0])
    for r in csv.DictReader(open(files[0])):
        grid = "x".join(str(r.get(k, "?")) for k in ("Grid_Size_X", "Grid_Size_Y", "Grid_Size_Z"))
        a = acc[(r["Kernel_Name"], grid)]
        a[0] += float(r["End_Timestamp"]) - float(r["Start_Timestamp"])
        a[1] += 1
    tot = sum(v[0] for v in acc.values())
    out.write(f"# {files[0]}\n{'kernel':80s} {'grid (threads)':>18s} {'calls':>7s} {'avg_us':>10s} {'total_ms':>10s} {'pct':>6s}\n")
    for (k, g), (ns, n) in sorted(acc.items(), key=lambda kv: -kv[1][0])[:40]:
        out.write(f"{short(k, 80):80s} {g:>18s} {n:7d} {ns / n / 1e3:10.2f} {ns / 1e6:10.2f} {100 * ns / tot:6.2f}\n")


def pmc(d, out):
    files = glob.glob(os.path.join(d, "**", "*counter_collection.csv"), recursive=True)
    if not files:
        out.write("no counter_collection.csv found\n")
        return
    acc = defaultdict(lambda: defaultdict(lambda: [0.0, 0]))
    for f in files:
        for r in csv.DictReader(open(f)):
            k = r.get("Kernel_Name") or r.get("Kernel Name")
            a = acc[k][r["Counter_Name"]]
            a[0] += float(r["Counter_Value"])
            a[1] += 1
    out.write(f"# {len(files)} counter file(s); per-dispatch means\n")
    ours = [k for k in acc if not (k.startswith("void at::") or "rocprim" in k or k.startswith("__amd_rocclr"))]
    for k in sorted(ours, key=lambda k: -max(v[0] for v in acc[k].values()))[:25]:
        out.write(short(k) + "\n")
        for c, (s, n) in acc[k].items():
            out.write(f"    {c:24s} mean {s / n:16.1f}   dispatches {n}\n")


if __name__ == "__main__":
    mode, d, dst = sys.argv[1], sys.argv[2], sys.argv[3]
    with open(dst, "w") as out:
        {"stats": kernel_stats, "shapes": kernel_shapes}.get(mode, pmc)(d, out)
    print(open(dst).read()[:6000])
