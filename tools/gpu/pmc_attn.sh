#!/bin/bash
# SQ counters of the prefill attention kernel at the bench's launch shape (4 sequences x 2142 tokens), two passes
REPO=${GRAFT_REPO_ROOT:-$(pwd)}
OUT=$REPO/gpurun_out/pmc_attn
rm -rf "$OUT"; mkdir -p "$OUT"
cd /tmp && export TMPDIR=/tmp
cat > /tmp/attn_mb.py <<PY
import sys, torch
sys.path.insert(0, "$REPO")
from aurora_amd import synthetic as S
from aurora_amd.engine import AuroraCapEngine, _rup
l = S.VICUNA_7B_16K
B, L0 = 4, 2142
eng = AuroraCapEngine({"vit": None, "llm": l}, {"llm": S.llm_weights(l)}, max_frames=1, max_batch=B, max_ctx=_rup(L0 + 64, 64), max_new_tokens=16)
eng.begin_batch(B, 16, None)
emb0 = (torch.randn(_rup(L0, 32), l["hidden_size"], device="cuda") * 0.02).half()
for b in range(B):
    eng.prefill(b, emb0.clone(), L0)
torch.cuda.synchronize()
eng.set_option("microbench_prefill_nseq", 4)
print("pre_attn us", eng.microbench("pre_attn", 40))
eng.close()
PY
pass() {
  local name=$1; shift
  timeout 600 rocprofv3 --pmc "$@" --kernel-trace --output-format csv -d "$OUT/$name" -o $name -- python /tmp/attn_mb.py > "$OUT/$name.log" 2>&1 || echo "pass $name failed"
}
pass a SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_WAIT_INST_LDS SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_LDS SQ_LDS_BANK_CONFLICT
pass b SQ_INSTS_VALU SQ_INSTS_LDS SQ_INSTS_SALU SQ_LDS_IDX_ACTIVE SQ_VALU_MFMA_BUSY_CYCLES SQ_BUSY_CYCLES SQ_WAVES SQ_INSTS_SMEM
python - <<PY
import csv, glob, collections
for name in ("a", "b"):
    for f in glob.glob("$OUT/%s/**/*counter_collection.csv" % name, recursive=True):
        acc = collections.defaultdict(lambda: [0.0, 0])
        for row in csv.DictReader(open(f)):
            k = row.get("Kernel_Name", "")
            if "attn_kernel" not in k or "decode" in k:
                continue
            key = (k[:60], row["Counter_Name"])
            acc[key][0] += float(row["Counter_Value"]); acc[key][1] += 1
        for (k, c), (v, n) in sorted(acc.items()):
            print(name, k, c, "mean per launch", round(v / n, 1), "n", n)
PY
find "$OUT" -name "*.csv" -size +1M -delete; find "$OUT" -name "*.db" -delete
tail -3 "$OUT/a.log"
