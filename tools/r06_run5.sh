set -u
R=$GRAFT_REPO_ROOT
O=$R/gpurun_out/r06_run5
mkdir -p $O
cd $R
python -m pytest tests -m gpu -x -q 2>&1 | tail -6 > $O/pytest_gpu.txt
OFFSIM_BENCH_TRACE=1 python bench.py --steps 3 --warmup 1 > $O/bench_default.json 2> $O/bench_default.err
cd /tmp && export TMPDIR=/tmp
rocprofv3 --pmc FETCH_SIZE --kernel-trace --output-format csv -d $O/calib_dma_u -- $R/tools/micro/calib_dma > $O/calib_dma_u.txt 2>&1
rocprofv3 --pmc FETCH_SIZE --kernel-trace --output-format csv -d $O/calib_dma_a -- $R/tools/micro/calib_dma aligned > $O/calib_dma_a.txt 2>&1
cd $O && python3 - <<'PY'
import csv, glob
for d in ("calib_dma_u", "calib_dma_a"):
    tot = 0.0
    for f in glob.glob(d + "/**/*counter_collection.csv", recursive=True):
        for r in csv.DictReader(open(f)):
            if "calib_dma" in r["Kernel_Name"] and r["Counter_Name"] == "FETCH_SIZE":
                tot += float(r["Counter_Value"])
    print(d, "FETCH_SIZE (KiB as reported)", tot, "-> bytes", tot * 1024)
PY
rm -rf $O/calib_dma_u $O/calib_dma_a
tail -3 $O/pytest_gpu.txt; grep "^\[bench" $O/bench_default.err | tail -40
