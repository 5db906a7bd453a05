#!/bin/bash
# Runs on the GPU box (via gpurun): kernel trace + the two HBM PMC passes (separate runs, as the microarchitecture guide
# prescribes) of the delayed leg -- tools/delayed_leg.py -- and the summary profiles/rNN/flush_pmc.txt is made from.
# usage: tools/delayed_pmc.sh <tag> [delayed_leg.py args...]
set -o pipefail
TAG=${1:-r04}; shift
ROOT=${GRAFT_REPO_ROOT:-/root/repo}
OUT=$ROOT/gpurun_out/dpmc_$TAG
mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp
rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/trace -- python3 $ROOT/tools/delayed_leg.py "$@" > $OUT/leg_trace.txt 2> $OUT/trace.err || { tail -5 $OUT/trace.err; exit 1; }
rocprofv3 --pmc FETCH_SIZE --output-format csv -d $OUT/pmc_fetch -- python3 $ROOT/tools/delayed_leg.py "$@" > $OUT/leg_fetch.txt 2> $OUT/pmc_fetch.err || { tail -5 $OUT/pmc_fetch.err; exit 1; }
rocprofv3 --pmc WRITE_SIZE --output-format csv -d $OUT/pmc_write -- python3 $ROOT/tools/delayed_leg.py "$@" > $OUT/leg_write.txt 2> $OUT/pmc_write.err || { tail -5 $OUT/pmc_write.err; exit 1; }
{
  echo "# python3 tools/delayed_leg.py $@"; cat $OUT/leg_trace.txt
  echo "# rocprofv3 --kernel-trace --stats (same command)"
  python3 - "$OUT" <<'PY'
import csv, glob, sys
f = glob.glob(sys.argv[1] + "/trace/**/*kernel_stats.csv", recursive=True)
for r in csv.DictReader(open(f[0])):
    if any(s in r["Name"] for s in ("flush", "gain_delayed", "k_predict", "panel")):
        print(f"{r['Name'].split('(')[0][:60]:60s} calls {r['Calls']:>5s} avg_us {float(r['AverageNs']) / 1e3:10.1f} min_us {float(r['MinNs']) / 1e3:10.1f} max_us {float(r['MaxNs']) / 1e3:10.1f}")
PY
  echo "# rocprofv3 --pmc FETCH_SIZE / WRITE_SIZE (separate passes, same command)"
  python3 $ROOT/tools/pmc_flush_summary.py $OUT/pmc_fetch $OUT/pmc_write
} > $OUT/flush_pmc.txt
cat $OUT/flush_pmc.txt
find $OUT -name '*.csv' -size +8M -delete
