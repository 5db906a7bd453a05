#!/bin/bash
# Runs on the GPU box (via gpurun): rocprofv3 kernel-trace stats + two PMC passes of the SAME bench command.
# usage: tools/profile_bench.sh <tag> [bench args...]
set -o pipefail
TAG=${1:-r01}; shift
ROOT=${GRAFT_REPO_ROOT:-/root/repo}
OUT=$ROOT/gpurun_out/prof_$TAG
mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp
ARGS="--no-cpu-baseline $@"
# kernel trace: the WHOLE bench line (every leg's kernels get a row); PMC passes: the contract leg only
rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/trace -- python3 $ROOT/bench.py $ARGS > $OUT/bench_trace.json 2> $OUT/trace.err || { tail -5 $OUT/trace.err; exit 1; }
rocprofv3 --pmc FETCH_SIZE --output-format csv -d $OUT/pmc_fetch -- python3 $ROOT/bench.py --only-main $ARGS > $OUT/bench_pmc_fetch.json 2> $OUT/pmc_fetch.err || { tail -5 $OUT/pmc_fetch.err; exit 1; }
rocprofv3 --pmc WRITE_SIZE --output-format csv -d $OUT/pmc_write -- python3 $ROOT/bench.py --only-main $ARGS > $OUT/bench_pmc_write.json 2> $OUT/pmc_write.err || { tail -5 $OUT/pmc_write.err; exit 1; }
python3 $ROOT/tools/summarize_profile.py $OUT > $OUT/summary.txt
cat $OUT/summary.txt
# keep only the small files (the per-dispatch CSVs can be large)
find $OUT -name '*.csv' -size +8M -delete
