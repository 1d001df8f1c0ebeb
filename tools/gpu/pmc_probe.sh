#!/bin/bash
# which kernel makes rocprofv3's counter mode crash?  short PMC passes with one knob changed at a time
REPO=${GRAFT_REPO_ROOT:-$(pwd)}
cd /tmp && export TMPDIR=/tmp
for v in 1 4; do
  rm -rf /tmp/pmcp
  timeout 600 rocprofv3 --pmc GRBM_GUI_ACTIVE --kernel-trace --output-format csv -d /tmp/pmcp -o p -- python $REPO/bench.py --no-cpu-baseline --no-power --batch ${PB:-128} --prefill-group 4 --steps 1 --warmup 0 --max_new_tokens 4 --no-graph --no-instrument --batch-mode --dec-attn-variant $v > /tmp/pmcp.log 2>&1
  echo "variant $v: rc=$? $(grep -c 'Segmentation\|SIGSEGV' /tmp/pmcp.log) segv lines; $(grep -o '"value": [0-9.]*' /tmp/pmcp.log | head -1)"
  grep "launch_\|enqueue" /tmp/pmcp.log | head -3
done
