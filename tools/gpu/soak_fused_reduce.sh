#!/bin/bash
# Root cause of round 3's fused split-K reduce failure, shown three ways (log -> profiles/r04_fused_reduce_soak.log):
#  1. tools/hazard_lab/store_hazard: an inline-asm dwordx4 store whose data registers are overwritten 0 / 1 / 2 / 4 wait states later.
#  2. the serving schedule (bench.py's id check of every cycle) with hand-over 2 of the AUR_LABS build = round 3's inline-asm partial stores:
#     expected to FAIL.
#  3. the same with hand-over 1 of the product build (raw-buffer builtins hipcc pads and counts), alternating with the two-launch form:
#     expected to pass every cycle; the pairs are also the throughput A/B.
# usage: tools/gpu/soak_fused_reduce.sh [lab_runs=2] [lab_steps=8] [pairs=3] [pair_steps=5]
LAB_RUNS=${1:-2}; LAB_STEPS=${2:-8}; PAIRS=${3:-3}; PAIR_STEPS=${4:-5}
mkdir -p gpurun_out
LOG=gpurun_out/r04_fused_reduce_soak.log
: > $LOG
echo "== 1. store hazard lab" >> $LOG
if [ -x tools/hazard_lab/store_hazard ]; then timeout 300 tools/hazard_lab/store_hazard >> $LOG 2>&1; else echo "(binary missing)" >> $LOG; fi
echo "== the hand-over unit test (product build)" >> $LOG
timeout 900 python -m pytest tests/test_gpu_skinny_lds.py -x -q -m gpu 2>&1 | grep -v amdgpu.ids | tail -3 >> $LOG
one() {   # $1 = label, $2 = fused value, $3 = steps, $4 = library ("" = product)
  local t0=$(date +%s)
  AURORA_HIP_SO=$4 timeout 1500 python bench.py --fused-reduce $2 --steps $3 --warmup 1 --no-cpu-baseline --no-instrument --no-power > gpurun_out/soak.json 2> gpurun_out/soak.err
  local rc=$?
  python - "$1" $rc $(( $(date +%s) - t0 )) >> $LOG 2>&1 <<'PY'
import json, sys
label, rc, secs = sys.argv[1], int(sys.argv[2]), sys.argv[3]
try:
    d = json.loads(open("gpurun_out/soak.json").read().strip().splitlines()[-1])
    print(f"{label}: rc {rc}, {d['steps']} cycles checked against the batch-mode ids, {d['value']:.3f} captions/s, ids crc {d['ids_checksum_rank0']}, {secs} s")
except Exception:
    err = [l for l in open("gpurun_out/soak.err").read().splitlines() if "amdgpu.ids" not in l]
    print(f"{label}: rc {rc} FAILED after {secs} s: " + " | ".join(err[-2:]))
PY
}
echo "== 2. serving schedule, round-3 inline-asm partial stores (AUR_LABS build, decode_fused_reduce 2)" >> $LOG
for i in $(seq $LAB_RUNS); do one "lab asm stores run $i" 2 $LAB_STEPS aurora_amd/libaurora_hip_labs.so; done
echo "== 3. serving schedule, product build: fused (1) / two-launch (0) alternating" >> $LOG
for i in $(seq $PAIRS); do
  one "fused reduce    pair $i" 1 $PAIR_STEPS ""
  one "two-launch form pair $i" 0 $PAIR_STEPS ""
done
cat $LOG
