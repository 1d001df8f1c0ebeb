#!/usr/bin/env python3
"""Where does a decode step go at few slots?  Reads a rocprofv3 --kernel-trace csv, takes the decode steps (the launches between two
argmax_advance kernels that hold no GEMM), and prints per kernel: launches per step, mean duration, mean idle gap in front of it.

    python tools/step_timeline.py <dir with *kernel_trace.csv> [out.txt]
"""
import csv
import glob
import os
import sys
from collections import OrderedDict


def short(name):
    n = name.split("(")[0]
    for p in ("void ", "_Z"):
        if n.startswith(p):
            n = n[len(p):]
    return n[:60]


def main():
    d = sys.argv[1]
    files = glob.glob(os.path.join(d, "**", "*kernel_trace.csv"), recursive=True)
    rows = []
    for f in files:
        with open(f) as fh:
            for r in csv.DictReader(fh):
                rows.append((int(r["Start_Timestamp"]), int(r["End_Timestamp"]), r.get("Kernel_Name") or r.get("Kernel Name")))
    rows.sort()
    # split at argmax_advance
    steps, cur = [], []
    for s, e, k in rows:
        cur.append((s, e, k))
        if "argmax_advance" in k:
            steps.append(cur)
            cur = []
    dec = [st for st in steps if 100 < len(st) < 400 and not any("gemm" in k for _, _, k in st)]
    if not dec:
        print("no decode steps found")
        return
    # keep steps that follow another decode step (the gap in front of the first kernel is then a step-to-step gap)
    agg = OrderedDict()
    tot_busy = tot_span = 0
    n = 0
    for i in range(1, len(steps)):
        st, prev = steps[i], steps[i - 1]
        if st not in dec or prev not in dec:
            continue
        n += 1
        last_end = prev[-1][1]
        tot_span += st[-1][1] - last_end
        for idx, (s, e, k) in enumerate(st):
            key = short(k)
            a = agg.setdefault(key, [0, 0.0, 0.0])
            a[0] += 1
            a[1] += e - s
            a[2] += max(0, s - last_end)
            tot_busy += e - s
            last_end = max(last_end, e)
    out = [f"# {n} decode steps (each preceded by another decode step); step span {tot_span / n / 1e3:.1f} us, kernels busy {tot_busy / n / 1e3:.1f} us "
           f"({100.0 * tot_busy / tot_span:.1f} %), idle between kernels {(tot_span - tot_busy) / n / 1e3:.1f} us",
           f"{'kernel':62s} {'per step':>8s} {'mean us':>9s} {'gap before us':>14s} {'step us':>9s} {'step gap us':>12s}"]
    for k, (c, du, gp) in agg.items():
        out.append(f"{k:62s} {c / n:8.1f} {du / c / 1e3:9.2f} {gp / c / 1e3:14.2f} {du / n / 1e3:9.1f} {gp / n / 1e3:12.1f}")
    txt = "\n".join(out)
    print(txt)
    if len(sys.argv) > 2:
        open(sys.argv[2], "w").write(txt + "\n")


if __name__ == "__main__":
    main()
