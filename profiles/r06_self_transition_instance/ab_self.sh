#!/bin/bash
# A/B of the scan's SELF instance (OFFSIM_ROWS_SELF = 0 / 1) on bench.py's workloads, row-packed kernel forced, and the window kernel beside it:
# tools/ab_self.sh OUTDIR      (GPU box, repo root)
out=gpurun_out/$1
mkdir -p $out
for wl in "cartpole 1000000" "iid 10000000" "grid 10000000" "cartpole 10000000"; do
  set -- $wl
  for self in 0 1; do
    OFFSIM_ROWS_SELF=$self timeout 900 python tools/diag_scan.py $1 $2 4096 rows 2>&1 | grep -v amdgpu.ids | sed "s/^/[$1 $2 rows SELF=$self] /" >> $out/ab_self.txt
  done
  timeout 900 python tools/diag_scan.py $1 $2 4096 win 2>&1 | grep -v amdgpu.ids | sed "s/^/[$1 $2 win] /" >> $out/ab_self.txt
done
cat $out/ab_self.txt
