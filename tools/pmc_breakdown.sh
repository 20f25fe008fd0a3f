#!/bin/bash
# FETCH_SIZE of the scan launch for builds in which parts of its traffic are stubbed (timing-only variants, profiles/r06_accept_traffic_experiments):
# usage (GPU box, repo root): tools/pmc_breakdown.sh OUTDIR lib1.so ... ("default" = product).  One rocprofv3 --pmc pass per build of
# tools/clock_rows.py 10000000 4096 (two resets + two scans); prints FETCH_SIZE summed over the k_eval_mc_rows dispatches.
out=$GRAFT_REPO_ROOT/gpurun_out/$1; shift
mkdir -p "$out"
cd /tmp && export TMPDIR=/tmp
for v in "$@"; do
  if [ "$v" = default ]; then unset OFFSIM_LIB; else export OFFSIM_LIB=$GRAFT_REPO_ROOT/rl-offline-simulation_amd/csrc/variants/$v; fi
  rocprofv3 --pmc FETCH_SIZE --kernel-trace --output-format csv -d $out/$v -- python3 $GRAFT_REPO_ROOT/tools/clock_rows.py 10000000 4096 > $out/$v.txt 2>&1
  python3 - "$out/$v" "$v" <<'PY'
import csv, glob, sys, collections
agg = collections.defaultdict(lambda: [0.0, 0])
for f in glob.glob(sys.argv[1] + "/**/*counter_collection.csv", recursive=True):
    for r in csv.DictReader(open(f)):
        if r["Counter_Name"] == "FETCH_SIZE":
            k = r["Kernel_Name"].split("(")[0][:48]
            agg[k][0] += float(r["Counter_Value"]); agg[k][1] += 1
for k, (v, n) in sorted(agg.items(), key=lambda x: -x[1][0])[:4]:
    print(sys.argv[2], k, "dispatches", n, "FETCH_SIZE bytes per dispatch %.4e" % (v * 1024 / n))
PY
  rm -rf $out/$v
done
