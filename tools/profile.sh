#!/bin/bash
# Runs on the GPU box (via gpurun): bench + rocprofv3 kernel trace + separate PMC passes.
# usage: tools/profile.sh <tag> [bench args...]
set -o pipefail
TAG=${1:-r01}; shift
R=${GRAFT_REPO_ROOT:-/root/repo}
OUT=$R/gpurun_out/$TAG
mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp
python3 $R/bench.py "$@" > $OUT/bench.json 2> $OUT/bench.err || { echo "bench failed"; tail -5 $OUT/bench.err; exit 1; }
echo "bench done"; cat $OUT/bench.json
rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/kt -- python3 $R/bench.py --no-cpu-baseline "$@" > $OUT/kt.log 2>&1 || { echo "kernel-trace failed"; tail -5 $OUT/kt.log; exit 1; }
echo "kernel trace done"
rocprofv3 --pmc FETCH_SIZE --output-format csv -d $OUT/pmc_fetch -- python3 $R/bench.py --no-cpu-baseline --steps 1 --warmup 1 "$@" > $OUT/pmc_fetch.log 2>&1 || { echo "pmc fetch failed"; tail -5 $OUT/pmc_fetch.log; exit 1; }
echo "pmc fetch done"
rocprofv3 --pmc WRITE_SIZE --output-format csv -d $OUT/pmc_write -- python3 $R/bench.py --no-cpu-baseline --steps 1 --warmup 1 "$@" > $OUT/pmc_write.log 2>&1 || { echo "pmc write failed"; tail -5 $OUT/pmc_write.log; exit 1; }
echo "pmc write done"
find $OUT -name "*.csv" | head -20
