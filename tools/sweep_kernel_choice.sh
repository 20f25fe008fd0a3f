#!/bin/bash
# Scan seconds per pass of the row-packed kernel (OFFSIM_SCAN_ROWS=1) and the window kernel (=0) over equal-state tables of 10 M rows,
# 1024 rollouts: the measurements behind BatchedPSRS._streams_apply's rule.  usage: tools/sweep_kernel_choice.sh > out.txt
cd "$(dirname "$0")/.."
run() {  # n_states n_actions
  for m in 1 0; do
    OFFSIM_SCAN_ROWS=$m python bench.py --n-states $1 --n-actions $2 --rollouts 1024 --no-configs --no-cpu-baseline --no-parity-check --steps 2 --warmup 1 2>/dev/null |
      python -c "import sys,json; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('states', $1, 'actions', $2, 'rows' if $m else 'win ', d['roofline']['kernel'], 'scan_s', round(d['scan_s_per_pass'],4), 'reset_s', round(d['reset_sampler_s_per_pass'],4), 'acceptance', round(d['acceptance'],3))"
  done
}
for s in 162 50 35 25 12 6; do run $s 2; done
for a in 3 4 5; do run 162 $a; done
run 50 5
run 25 3
