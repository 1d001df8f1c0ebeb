#!/bin/bash
# roctx ranges per stage: AURORA_ROCTX=1 makes the library push / pop a range around every stage's enqueue; rocprofv3
# --marker-trace records them beside the kernel trace.  Short run (cfg4, one step); summary -> gpurun_out/roctx_ranges.txt
R=${GRAFT_REPO_ROOT:-$(pwd)}
mkdir -p $R/gpurun_out
cd /tmp && export TMPDIR=/tmp
rm -rf /tmp/roctx_tr
AURORA_ROCTX=1 timeout 900 rocprofv3 --marker-trace --kernel-trace --output-format csv -d /tmp/roctx_tr -o t -- python $R/bench.py --config cfg4 --steps 1 --warmup 0 --no-cpu-baseline --no-instrument --no-power --decode-chunk 64 > /tmp/roctx_tr.log 2>&1
python - > $R/gpurun_out/roctx_ranges.txt <<'PY'
import csv, glob, collections
files = glob.glob("/tmp/roctx_tr/**/*marker*trace*.csv", recursive=True)
print("# rocprofv3 --marker-trace --kernel-trace -- python bench.py --config cfg4 --steps 1 (AURORA_ROCTX=1): roctx ranges pushed by libaurora_hip.so")
print("# files:", [f.split("/")[-1] for f in files])
acc = collections.defaultdict(lambda: [0, 0.0])
for f in files:
    for r in csv.DictReader(open(f)):
        name = r.get("Function") or r.get("Name") or r.get("Message") or str(r)
        try:
            d = float(r.get("End_Timestamp", 0)) - float(r.get("Start_Timestamp", 0))
        except ValueError:
            d = 0.0
        acc[name][0] += 1
        acc[name][1] += d
print(f"{'range':24s} {'count':>8s} {'host enqueue time, ms':>24s}")
for k, (n, d) in sorted(acc.items(), key=lambda kv: -kv[1][1]):
    print(f"{k[:24]:24s} {n:8d} {d / 1e6:24.2f}")
PY
tail -2 /tmp/roctx_tr.log | cut -c1-200 >> $R/gpurun_out/roctx_ranges.txt
cat $R/gpurun_out/roctx_ranges.txt
