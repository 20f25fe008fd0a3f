#!/bin/bash
# reset_sampler seconds (tools/time_reset.py, keyed, headline size) for several builds on one box: tools/ab_reset.sh OUTFILE lib1.so ... ("default" = product)
out=$1; shift
: > "$out"
for rep in 1 2; do
for v in "$@"; do
  if [ "$v" = default ]; then unset OFFSIM_LIB; else export OFFSIM_LIB=$PWD/rl-offline-simulation_amd/csrc/variants/$v; fi
  echo "$v: $(timeout 300 python tools/time_reset.py 10000000 162 4096 keyed 2>&1 | tail -1)" >> "$out"
done
done
cat "$out"
