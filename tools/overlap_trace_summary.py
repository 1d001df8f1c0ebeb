"""Summarise a rocprofv3 --kernel-trace of the overlapped schedule: which kernels ran on which queue, their mean duration, and how
much of the wall time had kernels of the decode queue(s) and of the front-end queue in flight at the same moment.
  python tools/overlap_trace_summary.py <dir with *_kernel_trace.csv> [out.txt]"""
import csv
import glob
import os
import sys
from collections import defaultdict


def main():
    d = sys.argv[1]
    out = open(sys.argv[2], "w") if len(sys.argv) > 2 else sys.stdout
    files = glob.glob(os.path.join(d, "**", "*kernel_trace.csv"), recursive=True)
    if not files:
        print("no kernel_trace.csv under", d, file=out)
        return
    rows = []
    with open(files[0]) as f:
        for r in csv.DictReader(f):
            rows.append((r["Kernel_Name"], int(r["Queue_Id"]), int(r["Start_Timestamp"]), int(r["End_Timestamp"])))
    rows.sort(key=lambda r: r[2])
    front_names = ("gemm256_kernel", "attn_kernel", "tome_", "norm_kernel", "gemm_kernel", "im2col", "vit_", "splice", "gather")
    # the timed region: from the first staged front end onwards is hard to tell apart; use the last 60 % of the trace (steady state)
    t0, t1 = rows[0][2], rows[-1][3]
    lo = t0 + int(0.4 * (t1 - t0))
    rows = [r for r in rows if r[2] >= lo]
    by_q = defaultdict(lambda: [0, 0])
    for name, q, s, e in rows:
        by_q[q][0] += 1
        by_q[q][1] += e - s
    print("# steady-state part of the trace: %.1f ms, %d kernels" % ((t1 - lo) / 1e6, len(rows)), file=out)
    for q, (n, busy) in sorted(by_q.items()):
        print("queue %d: %d kernels, busy %.1f ms (%.0f %% of the window)" % (q, n, busy / 1e6, 100.0 * busy / (t1 - lo)), file=out)
    # queue roles: the queue with most gemm256 time is the front end
    gq = defaultdict(int)
    for name, q, s, e in rows:
        if "gemm256_kernel" in name:
            gq[q] += e - s
    fq = max(gq, key=gq.get) if gq else None
    print("front-end queue: %s" % fq, file=out)
    # sweep: time with (front-end kernel in flight) AND (a kernel of another queue in flight)
    ev = []
    for name, q, s, e in rows:
        k = 0 if q == fq else 1
        ev.append((s, 1, k))
        ev.append((e, -1, k))
    ev.sort()
    act = [0, 0]
    last = ev[0][0]
    both = only_f = only_d = idle = 0
    for t, dlt, k in ev:
        span = t - last
        if act[0] > 0 and act[1] > 0:
            both += span
        elif act[0] > 0:
            only_f += span
        elif act[1] > 0:
            only_d += span
        else:
            idle += span
        act[k] += dlt
        last = t
    tot = both + only_f + only_d + idle
    print("both queues busy %.1f %%, only decode %.1f %%, only front end %.1f %%, idle %.1f %% of %.1f ms" %
          (100.0 * both / tot, 100.0 * only_d / tot, 100.0 * only_f / tot, 100.0 * idle / tot, tot / 1e6), file=out)
    # per kernel: mean duration when it overlapped a kernel of the other role for more than half of its span vs when it ran alone
    spans = {0: sorted((s, e) for n, q, s, e in rows if q == fq), 1: sorted((s, e) for n, q, s, e in rows if q != fq)}
    import bisect
    starts = {k: [s for s, e in v] for k, v in spans.items()}

    def overlap_frac(k_other, s, e):
        v, st = spans[k_other], starts[k_other]
        i = max(bisect.bisect_right(st, s) - 1, 0)
        cov = 0
        while i < len(v) and v[i][0] < e:
            cov += max(0, min(e, v[i][1]) - max(s, v[i][0]))
            i += 1
        return cov / max(e - s, 1)

    agg = defaultdict(lambda: [0, 0.0, 0, 0.0])
    for name, q, s, e in rows:
        k = 0 if q == fq else 1
        fr = overlap_frac(1 - k, s, e)
        a = agg[(k, name.split("(")[0][:70])]
        if fr > 0.5:
            a[0] += 1
            a[1] += e - s
        elif fr < 0.05:
            a[2] += 1
            a[3] += e - s
    print("%-8s %-70s %8s %10s %8s %10s" % ("role", "kernel", "n_ovl", "us_ovl", "n_alone", "us_alone"), file=out)
    for (k, name), a in sorted(agg.items(), key=lambda kv: -(kv[1][1] + kv[1][3]))[:24]:
        print("%-8s %-70s %8d %10.1f %8d %10.1f" % ("front" if k == 0 else "decode", name, a[0], a[1] / max(a[0], 1) / 1e3, a[2], a[3] / max(a[2], 1) / 1e3), file=out)


if __name__ == "__main__":
    main()
