#!/bin/bash
# scan seconds of the row-packed kernel by launch shape: chain wavefronts per workgroup (OFFSIM_ROWS_WAVES) x rollouts per chain wavefront
# (OFFSIM_ROWS_PER_WAVE); "auto auto" = the launcher's choice, "4 4" = packed sixteen rollouts to a CU (every launch before round 4's second half)
#   tools/sweep_launch_shape.sh OUTFILE ["R waves rows" ...]  -> one line per (rollouts, shape)
out=$1; shift
: > "$out"
B="--steps 2 --warmup 1 --no-cpu-baseline --no-configs"
if [ $# -eq 0 ]; then
  set -- "128 auto auto" "256 auto auto" "256 4 4" "512 auto auto" "512 1 4" "512 4 4" "1024 auto auto" "1024 1 4" "1024 4 4" "2048 auto auto" "2048 4 2" "2048 4 4" "3072 auto auto" "3072 4 4" "4096 auto auto"
fi
for cfg in "$@"; do
  set -- $cfg
  if [ "$2" = auto ]; then unset OFFSIM_ROWS_WAVES; else export OFFSIM_ROWS_WAVES=$2; fi
  if [ "$3" = auto ]; then unset OFFSIM_ROWS_PER_WAVE; else export OFFSIM_ROWS_PER_WAVE=$3; fi
  line=$(timeout 300 python bench.py --rollouts $1 $B 2>/dev/null | tail -1)
  echo "rollouts $1 waves $2 rows/wave $3: $(echo "$line" | python -c "import sys,json; d=json.loads(sys.stdin.read()); print('scan', round(d['scan_s_per_pass'],4), 'reset', round(d['reset_sampler_s_per_pass'],4), 'value', '%.3e'%d['value'], 'parity', d.get('parity_check',{}).get('ok'))")" >> "$out"
done
cat "$out"
