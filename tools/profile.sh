#!/bin/bash
# Runs on the GPU box (via gpurun): bench + rocprofv3 kernel trace + separate PMC passes (never --pmc together with a trace).
# usage: tools/profile.sh <tag> [bench args...]     -> gpurun_out/<tag>/{bench.json,kt/,pmc_fetch/,pmc_write/,hbm.txt}
set -o pipefail
TAG=${1:-r02}; shift
R=${GRAFT_REPO_ROOT:-/root/repo}
OUT=$R/gpurun_out/$TAG
mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp
python3 $R/bench.py "$@" > $OUT/bench.json 2> $OUT/bench.err || { echo "bench failed"; tail -5 $OUT/bench.err; exit 1; }
echo "bench done"; cat $OUT/bench.json
rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/kt -- python3 $R/bench.py --no-cpu-baseline --no-check "$@" > $OUT/kt.log 2>&1 || { echo "kernel-trace failed"; tail -5 $OUT/kt.log; exit 1; }
echo "kernel trace done"
for c in FETCH_SIZE WRITE_SIZE; do
  rocprofv3 --pmc $c --output-format csv -d $OUT/pmc_$c -- python3 $R/bench.py --no-cpu-baseline --no-check "$@" --steps 1 --warmup 1 > $OUT/pmc_$c.log 2>&1 || { echo "pmc $c failed"; tail -5 $OUT/pmc_$c.log; exit 1; }
done
python3 - <<PY | tee $OUT/hbm.txt
import csv, glob, collections, re
acc = collections.defaultdict(float); n = collections.defaultdict(int)
for c in ("FETCH_SIZE", "WRITE_SIZE"):
    for f in glob.glob("$OUT/pmc_%s/*/*counter_collection.csv" % c):
        for r in csv.DictReader(open(f)):
            if re.search(r"rtiow_wave_kernel<\d+, \d+, false", r["Kernel_Name"]):  # the TIMED (counter-free) instantiation
                acc[r["Counter_Name"]] += float(r["Counter_Value"]); n[r["Counter_Name"]] += 1
# MI355X_MICROARCH.md §HBM: both counters are in KiB on gfx950 and FETCH_SIZE reports half of the bytes fetched
fetch, write = acc["FETCH_SIZE"], acc["WRITE_SIZE"]
print("timed-kernel launches seen:", dict(n))
print("FETCH_SIZE_KiB", fetch, "WRITE_SIZE_KiB", write)
steps = max(1, n["FETCH_SIZE"] // 2)  # two launches per step (8-sample cost probe + cost-sorted remainder)
print("steps seen", steps, "hbm_bytes_per_step", (2.0 * fetch + write) * 1024.0 / steps, "hbm_bytes_per_launch", (2.0 * fetch + write) * 1024.0 / max(1, n["FETCH_SIZE"]))
PY
find $OUT/kt -name "*kernel_stats.csv" | head -3
