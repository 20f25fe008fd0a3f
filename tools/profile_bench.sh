#!/bin/bash
# Run on the GPU box (via gpurun) from the repo root: rocprofv3 kernel-trace stats of the default bench command, then
# separate PMC passes (memory-side counters, then SQ counters) of the same command.  usage: profile_bench.sh <out-name> [bench args]
# The summary (with the digest of the kernel sources it measured) lands in gpurun_out/<out-name>/summary.json; copy it,
# kernel_stats.csv and bench_under_rocprof.json into profiles/<round>_... to have bench.py report `roofline.traffic` from it.
set -u
NAME=${1:-prof}
shift || true
OUT=$GRAFT_REPO_ROOT/gpurun_out/$NAME
mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp
ARGS="--steps 1 --warmup 1 --no-cpu-baseline --no-parity-check --no-configs $*"
DIGEST=$(python3 $GRAFT_REPO_ROOT/bench.py --print-csrc-digest)
rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/trace -- python3 $GRAFT_REPO_ROOT/bench.py $ARGS > $OUT/bench_under_rocprof.json 2> $OUT/trace.err
rocprofv3 --pmc FETCH_SIZE --kernel-trace --output-format csv -d $OUT/pmc_fetch -- python3 $GRAFT_REPO_ROOT/bench.py $ARGS > /dev/null 2> $OUT/pmc_fetch.err
rocprofv3 --pmc WRITE_SIZE --kernel-trace --output-format csv -d $OUT/pmc_write -- python3 $GRAFT_REPO_ROOT/bench.py $ARGS > /dev/null 2> $OUT/pmc_write.err
rocprofv3 --pmc SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_INSTS_VMEM SQ_INSTS_BRANCH SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_ACTIVE_INST_ANY --kernel-trace --output-format csv -d $OUT/pmc_sq -- python3 $GRAFT_REPO_ROOT/bench.py $ARGS > /dev/null 2> $OUT/pmc_sq.err
cd $OUT && python3 - "$DIGEST" "$ARGS" <<'PY'
import csv, glob, collections, json, sys
def kernel_stats(d):
    rows = []
    for f in glob.glob(d + "/**/*kernel_stats.csv", recursive=True):
        rows += list(csv.DictReader(open(f)))
    return rows
def pmc(d):
    agg = collections.defaultdict(lambda: collections.defaultdict(float))
    for f in glob.glob(d + "/**/*counter_collection.csv", recursive=True):
        for r in csv.DictReader(open(f)):
            k = r["Kernel_Name"].split("(")[0][:60]
            agg[k][r["Counter_Name"]] += float(r["Counter_Value"])
    return {k: dict(v) for k, v in agg.items()}
summary = {"csrc_digest": sys.argv[1], "bench_args": sys.argv[2], "kernel_stats": kernel_stats("trace"), "pmc_fetch": pmc("pmc_fetch"),
           "pmc_write": pmc("pmc_write"), "pmc_sq": pmc("pmc_sq")}
json.dump(summary, open("summary.json", "w"), indent=1)
with open("kernel_stats.csv", "w") as f:
    rows = summary["kernel_stats"]
    if rows:
        w = csv.DictWriter(f, fieldnames=list(rows[0].keys())); w.writeheader(); w.writerows(rows)
for r in summary["kernel_stats"][:6]:
    print({k: r[k] for k in ("Name", "Calls", "AverageNs", "Percentage")})
for name in ("pmc_fetch", "pmc_write"):
    for k, v in summary[name].items():
        if "eval_mc" in k or "shuffle" in k:
            print(name, k, v)
PY
# keep the merged-back payload small: the raw per-dispatch CSVs are large
find $OUT -name "*counter_collection.csv" -size +1M -delete
find $OUT -name "*kernel_trace.csv" -size +1M -delete
rm -rf $OUT/trace $OUT/pmc_fetch $OUT/pmc_write $OUT/pmc_sq 2>/dev/null
ls -la $OUT
