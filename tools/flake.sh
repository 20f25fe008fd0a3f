#!/bin/bash
# usage: tools/flake.sh N test-selector [lib]: runs a GPU test N times, prints the number of failures
n=$1; sel=$2; lib=${3:-}
if [ -n "$lib" ]; then export OFFSIM_LIB=$PWD/rl-offline-simulation_amd/csrc/variants/$lib; fi
fail=0
for i in $(seq $n); do
  python -m pytest tests/test_gpu_round2.py -m gpu -x -q -k "$sel" > /tmp/flake.log 2>&1 || { fail=$((fail+1)); grep -E "^E |FAIL|assert" /tmp/flake.log | head -5; }
done
echo "lib=${lib:-default} failures=$fail of $n"
