#!/bin/bash
# clock_rows.py (product build: cycles per iteration, clock, spread) for several builds on one box: tools/ab_clock.sh OUTDIR lib1.so ... ("default" = product)
out=$1; shift
mkdir -p "$out"
for v in "$@"; do
  if [ "$v" = default ]; then unset OFFSIM_LIB; else export OFFSIM_LIB=$PWD/rl-offline-simulation_amd/csrc/variants/$v; fi
  case "$v" in
    *prof*) timeout 600 python tools/prof_rows.py 10000000 4096 > "$out/$v.txt" 2>&1 ;;
    *) timeout 600 python tools/clock_rows.py 10000000 4096 > "$out/$v.txt" 2>&1 ;;
  esac
  echo "$v rc=$?"
done
