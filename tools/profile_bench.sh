#!/bin/bash
# Run on the GPU box (via gpurun) from the repo root: rocprofv3 kernel-trace stats of the default bench command,
# then separate PMC passes for the memory-side counters of the same command.  Summaries land in gpurun_out/prof_r1/.
set -u
OUT=$GRAFT_REPO_ROOT/gpurun_out/prof_r1
mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp
ARGS="--steps 1 --warmup 1 --no-cpu-baseline"
rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/trace -- python3 $GRAFT_REPO_ROOT/bench.py $ARGS > $OUT/bench_under_rocprof.json 2> $OUT/trace.err
rocprofv3 --pmc FETCH_SIZE --kernel-trace --output-format csv -d $OUT/pmc_fetch -- python3 $GRAFT_REPO_ROOT/bench.py $ARGS > /dev/null 2> $OUT/pmc_fetch.err
rocprofv3 --pmc WRITE_SIZE --kernel-trace --output-format csv -d $OUT/pmc_write -- python3 $GRAFT_REPO_ROOT/bench.py $ARGS > /dev/null 2> $OUT/pmc_write.err
rocprofv3 --pmc SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_INSTS_VMEM SQ_INSTS_BRANCH SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_ACTIVE_INST_ANY --kernel-trace --output-format csv -d $OUT/pmc_sq -- python3 $GRAFT_REPO_ROOT/bench.py $ARGS > /dev/null 2> $OUT/pmc_sq.err
cd $OUT && python3 - <<'PY'
import csv, glob, collections, json
def kernel_stats(d):
    rows = []
    for f in glob.glob(d + "/**/*kernel_stats.csv", recursive=True):
        rows += list(csv.DictReader(open(f)))
    return rows
def pmc(d):
    agg = collections.defaultdict(lambda: collections.defaultdict(float)); n = collections.Counter()
    for f in glob.glob(d + "/**/*counter_collection.csv", recursive=True):
        for r in csv.DictReader(open(f)):
            k = r["Kernel_Name"].split("(")[0][:60]
            agg[k][r["Counter_Name"]] += float(r["Counter_Value"])
    return {k: dict(v) for k, v in agg.items()}
summary = {"kernel_stats": kernel_stats("trace"), "pmc_fetch": pmc("pmc_fetch"), "pmc_write": pmc("pmc_write"), "pmc_sq": pmc("pmc_sq")}
json.dump(summary, open("summary.json", "w"), indent=1)
for r in summary["kernel_stats"][:12]:
    print(r)
PY
# keep the merged-back payload small: the raw per-dispatch CSVs of the PMC passes are large
find $OUT -name "*counter_collection.csv" -size +2M -delete
find $OUT -name "*kernel_trace.csv" -size +2M -delete
ls -la $OUT
