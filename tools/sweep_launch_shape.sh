#!/bin/bash
# scan seconds of the row-packed kernel by launch shape (chain wavefronts per workgroup, OFFSIM_ROWS_WAVES; "auto" = the launcher's choice):
#   tools/sweep_launch_shape.sh OUTFILE  -> one line per (rollouts, shape)
out=$1
: > "$out"
B="--steps 2 --warmup 1 --no-cpu-baseline --no-configs"
for cfg in "512 auto" "512 4" "512 2" "1024 auto" "1024 4" "2048 auto" "2048 4" "3072 auto" "3072 4" "4096 auto"; do
  set -- $cfg
  if [ "$2" = auto ]; then unset OFFSIM_ROWS_WAVES; else export OFFSIM_ROWS_WAVES=$2; fi
  line=$(timeout 300 python bench.py --rollouts $1 $B 2>/dev/null | tail -1)
  echo "rollouts $1 waves $2: $(echo "$line" | python -c "import sys,json; d=json.loads(sys.stdin.read()); print('scan', round(d['scan_s_per_pass'],4), 'reset', round(d['reset_sampler_s_per_pass'],4), 'value', '%.3e'%d['value'], 'parity', d.get('parity_check',{}).get('ok'))")" >> "$out"
done
cat "$out"
