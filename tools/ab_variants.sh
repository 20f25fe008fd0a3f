#!/bin/bash
# A/B timing of builds of the same sources on one box: usage  tools/ab_variants.sh OUTDIR lib1.so lib2.so ...  (paths relative to csrc/variants/,
# "default" = the product library).  Every variant runs the default bench (10 M x 4096, parity check against the oracle on four seeds).
out=$1; shift
mkdir -p "$out"
for v in "$@"; do
  if [ "$v" = default ]; then unset OFFSIM_LIB; else export OFFSIM_LIB=$PWD/rl-offline-simulation_amd/csrc/variants/$v; fi
  case "$v" in
    *prof*) timeout 600 python tools/prof_rows.py 10000000 4096 > "$out/$v.prof.txt" 2> "$out/$v.err" ;;
    *) timeout 900 python bench.py --steps 2 --warmup 1 --no-cpu-baseline > "$out/$v.json" 2> "$out/$v.err" ;;
  esac
  echo "$v rc=$?"
done
