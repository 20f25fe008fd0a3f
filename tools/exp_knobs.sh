#!/bin/bash
out=gpurun_out/r05_knobs4.txt; : > $out
run() { echo "== $1" >> $out; shift; env "$@" python tools/clock_rows.py 10000000 4096 2>&1 | grep -v amdgpu.ids | sed -n '1,4p' >> $out; }
V=$PWD/rl-offline-simulation_amd/csrc/variants
run "default" X=1
run "no reward loads (timing only)" OFFSIM_LIB=$V/lib_norw.so
cat $out
