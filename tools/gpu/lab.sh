#!/bin/bash
# One driver for the GPU-side lab runs of a round (replaces the per-experiment run_*.sh files).  Each task appends to gpurun_out/<task>.log;
# summaries worth keeping are copied to profiles/ by hand.
#   tools/gpu/lab.sh <task> [args] [-- <task> [args] ...]
# tasks:
#   tests [files...]        pytest -m gpu, one process per file (default: every tests/test_gpu_*.py)
#   gemm                    tools/gemm_probe.py (shapes x both tile kernels, K sweep)
#   micro <batch> [kernels] tools/microbench.py at <batch> slots
#   bench <name> [flags]    python bench.py [flags] -> gpurun_out/bench_<name>.json (+ one line in bench.log)
#   ab <flagA> <flagB> <pairs> [flags]   alternating bench pairs, e.g. ab "--fused-reduce 1" "--fused-reduce 0" 3 --steps 3
mkdir -p gpurun_out
summ() { python - "$1" "$2" <<'PY'
import json, sys
try:
    d = json.loads(open(sys.argv[1]).read().strip().splitlines()[-1])
    pw = d.get("power") or {}
    print(f"{sys.argv[2]}: {d['value']:.3f} captions/s, {d['ms_per_step']:.1f} ms/step, p50 TTFT {d.get('p50_ttft_ms')}, single clip {d.get('ttft_ms_single_clip')}, "
          f"{pw.get('socket_power_w_p50')} W, sclk {pw.get('sclk_mhz_p50')}, step frac {(d.get('roofline_step') or {}).get('frac')}")
except Exception as e:
    print(f"{sys.argv[2]}: FAILED ({e})")
PY
}
while [ $# -gt 0 ]; do
  task=$1; shift
  args=()
  while [ $# -gt 0 ] && [ "$1" != "--" ]; do args+=("$1"); shift; done
  [ "$1" == "--" ] && shift
  case $task in
    tests)
      files=("${args[@]}"); [ ${#files[@]} -eq 0 ] && files=(tests/test_gpu_*.py)
      for f in "${files[@]}"; do
        echo "=== $f" >> gpurun_out/tests.log
        timeout 1500 python -m pytest $f -q -m gpu -x --no-header -p no:cacheprovider 2>&1 | grep -v amdgpu.ids | tail -25 >> gpurun_out/tests.log
      done
      grep -E "^===|passed|failed|error" gpurun_out/tests.log | tail -40 ;;
    gemm)
      timeout 900 python tools/gemm_probe.py 2>&1 | grep -v amdgpu.ids >> gpurun_out/gemm.log
      tail -80 gpurun_out/gemm.log ;;
    micro)
      b=${args[0]}; only=${args[1]:-}
      timeout 900 python tools/microbench.py --batch $b ${only:+--only $only} "${args[@]:2}" 2>&1 | grep -v amdgpu.ids >> gpurun_out/micro.log
      tail -30 gpurun_out/micro.log ;;
    bench)
      name=${args[0]}
      timeout 2400 python bench.py "${args[@]:1}" > gpurun_out/bench_$name.json 2> gpurun_out/bench_$name.err
      summ gpurun_out/bench_$name.json "bench $name (${args[*]:1})" | tee -a gpurun_out/bench.log ;;
    ab)
      fa=${args[0]}; fb=${args[1]}; n=${args[2]}
      for i in $(seq $n); do
        for f in "$fa" "$fb"; do
          timeout 2400 python bench.py $f "${args[@]:3}" --no-cpu-baseline --no-instrument > gpurun_out/ab.json 2> gpurun_out/ab.err
          summ gpurun_out/ab.json "pair $i [$f]" | tee -a gpurun_out/ab.log
        done
      done ;;
    *) echo "unknown task $task"; exit 2 ;;
  esac
done
