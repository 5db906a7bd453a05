#!/bin/bash
# usage (on the GPU box): tools/single_pmc.sh  -> gpurun_out/single_pmc/summary.txt
set -o pipefail
ROOT=${GRAFT_REPO_ROOT:-/root/repo}
OUT=$ROOT/gpurun_out/single_pmc
mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp
rocprofv3 --pmc FETCH_SIZE --output-format csv -d $OUT/fetch -- python3 $ROOT/tools/single_pmc.py 1 > $OUT/run_fetch.txt 2> $OUT/fetch.err || { tail -5 $OUT/fetch.err; exit 1; }
rocprofv3 --pmc WRITE_SIZE --output-format csv -d $OUT/write -- python3 $ROOT/tools/single_pmc.py 1 > $OUT/run_write.txt 2> $OUT/write.err || { tail -5 $OUT/write.err; exit 1; }
python3 - <<PY > $OUT/summary.txt
import csv, glob, os
def avg(sub, name):
    f = max(glob.glob(os.path.join("$OUT", sub, "**", "*counter_collection.csv"), recursive=True), key=os.path.getmtime)
    v = [float(r["Counter_Value"]) for r in csv.DictReader(open(f)) if r["Counter_Name"] == name and "k_correct_fused" in r["Kernel_Name"]]
    return sum(v) / len(v), len(v)
fs, n1 = avg("fetch", "FETCH_SIZE"); ws, n2 = avg("write", "WRITE_SIZE")
alg = 16 * 2003 ** 2
print(open("$OUT/run_fetch.txt").read().strip())
print(f"k_correct_fused, single filter n=1000: FETCH_SIZE avg {fs:.0f} KiB over {n1} launches -> {2 * fs * 1024 / 1e6:.2f} MB (x2 correction); "
      f"WRITE_SIZE avg {ws:.0f} KiB -> {ws * 1024 / 1e6:.2f} MB; HBM traffic {(2 * fs + ws) * 1024 / 1e6:.2f} MB per correction vs algorithmic {alg / 1e6:.2f} MB")
PY
cat $OUT/summary.txt
