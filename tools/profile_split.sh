set -u
OUT=$GRAFT_REPO_ROOT/gpurun_out/prof_split
mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp
export OFFSIM_SCAN_SPLIT=1
rocprofv3 --pmc SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_INSTS_VMEM SQ_INSTS_BRANCH SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_ACTIVE_INST_ANY --kernel-trace --output-format csv -d $OUT/pmc_sq -- python3 $GRAFT_REPO_ROOT/bench.py --steps 1 --warmup 1 --no-cpu-baseline > /dev/null 2> $OUT/pmc_sq.err
python3 - <<'PY'
import csv, glob, collections
agg = collections.defaultdict(lambda: collections.defaultdict(float))
for f in glob.glob("/tmp/../" + __import__("os").environ["GRAFT_REPO_ROOT"] + "/gpurun_out/prof_split/pmc_sq/**/*counter_collection.csv", recursive=True):
    for r in csv.DictReader(open(f)):
        agg[r["Kernel_Name"].split("(")[0][:50]][r["Counter_Name"]] += float(r["Counter_Value"])
for k, v in agg.items():
    if "k_eval_mc" in k: print(k, {a: f"{b:.4g}" for a, b in v.items()})
PY
find $OUT -name "*counter_collection.csv" -size +2M -delete
find $OUT -name "*kernel_trace.csv" -size +2M -delete
