#!/bin/bash
# Quick SQ-counter pass of the default bench (run on the GPU box via gpurun): per-kernel instruction mix and wait cycles.
set -u
OUT=$GRAFT_REPO_ROOT/gpurun_out/${1:-prof_quick}
mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp
ARGS="--steps 1 --warmup 0 --no-cpu-baseline --no-parity-check ${2:-}"
rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/trace -- python3 $GRAFT_REPO_ROOT/bench.py $ARGS > $OUT/bench.json 2> $OUT/trace.err
rocprofv3 --pmc SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_INSTS_VMEM SQ_INSTS_BRANCH SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_ACTIVE_INST_ANY --kernel-trace --output-format csv -d $OUT/pmc_sq -- python3 $GRAFT_REPO_ROOT/bench.py $ARGS > /dev/null 2> $OUT/pmc_sq.err
rocprofv3 --pmc SQ_WAIT_INST_ANY SQ_WAIT_INST_LDS SQ_INSTS_SMEM SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_LDS SQ_ACTIVE_INST_SCA SQ_ACTIVE_INST_VMEM SQ_LDS_BANK_CONFLICT --kernel-trace --output-format csv -d $OUT/pmc_sq2 -- python3 $GRAFT_REPO_ROOT/bench.py $ARGS > /dev/null 2> $OUT/pmc_sq2.err
cd $OUT && python3 - <<'PY'
import csv, glob, collections, json
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
summary = {"kernel_stats": kernel_stats("trace"), "pmc_sq": pmc("pmc_sq"), "pmc_sq2": pmc("pmc_sq2")}
json.dump(summary, open("summary.json", "w"), indent=1)
for r in summary["kernel_stats"][:8]:
    print({k: r[k] for k in ("Name", "Calls", "TotalDurationNs", "AverageNs", "Percentage")})
for name in ("pmc_sq", "pmc_sq2"):
    for k, v in summary[name].items():
        if "eval_mc" in k or "shuffle" in k:
            print(name, k, {a: f"{b:.4g}" for a, b in v.items()})
PY
find $OUT -name "*counter_collection.csv" -size +2M -delete
find $OUT -name "*kernel_trace.csv" -size +2M -delete
cat $OUT/bench.json | head -c 600
