#!/bin/bash
# SQ counters of gemm256_kernel on an epilogue-dominated shape (K = 128) and a K-loop-dominated one (K = 5120): where do a wave's cycles go?
# Separate --pmc passes with --kernel-trace only (gpurun refuses other trace domains beside counters).  Summary -> gpurun_out/pmc_gemm.txt
R=${GRAFT_REPO_ROOT:-$(pwd)}
mkdir -p $R/gpurun_out
cd /tmp && export TMPDIR=/tmp
OUT=$R/gpurun_out/pmc_gemm.txt
: > $OUT
for shape in "81920 128 5120" "81920 5120 5120"; do
  i=0
  for set in "SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_LDS SQ_WAIT_INST_LDS SQ_ACTIVE_INST_SCA" \
             "SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_INSTS_VMEM_WR SQ_INSTS_VMEM_RD SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_WAVES" \
             "SQ_ACTIVE_INST_VMEM SQ_ACTIVE_INST_MISC SQ_INSTS_BRANCH SQ_INST_CYCLES_SALU SQ_VALU_MFMA_BUSY_CYCLES SQ_BUSY_CYCLES SQ_IFETCH SQ_WAIT_INST_ANY"; do
    i=$((i+1))
    rm -rf /tmp/pg_$i
    timeout 600 rocprofv3 --pmc $set --kernel-trace --output-format csv -d /tmp/pg_$i -o p -- python $R/tools/gemm_lab/k_probe.py $shape 6 > /tmp/pg_$i.log 2>&1 || echo "pass $i failed: $(tail -2 /tmp/pg_$i.log)" >> $OUT
    python - "$shape" /tmp/pg_$i >> $OUT <<'PY'
import csv, glob, sys
from collections import defaultdict
acc = defaultdict(lambda: [0.0, 0])
for f in glob.glob(sys.argv[2] + "/**/*counter_collection.csv", recursive=True):
    for r in csv.DictReader(open(f)):
        k = r.get("Kernel_Name") or r.get("Kernel Name")
        if k and "gemm256_kernel" in k:
            a = acc[r["Counter_Name"]]
            a[0] += float(r["Counter_Value"]); a[1] += 1
print(f"shape {sys.argv[1]}: " + "  ".join(f"{c} {v[0] / max(v[1], 1):.4g}" for c, v in sorted(acc.items())))
PY
  done
done
cat $OUT
