#!/bin/bash
# L1 / L2 hit-rate counters for a non-headline config. usage: tools/pmc_cache.sh <tag> <cfg4|cfg5>
TAG=$1; CFG=$2
R=${GRAFT_REPO_ROOT:-/root/repo}
OUT=$R/gpurun_out/$TAG; mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp
export CFG4_SPP=4 CFG5_SPP=1
i=0
for set in "TCP_TOTAL_CACHE_ACCESSES_sum TCP_TCC_READ_REQ_sum" "TCC_HIT_sum TCC_MISS_sum" "TCC_EA0_RDREQ_sum TCC_REQ_sum"; do
  i=$((i+1))
  rocprofv3 --pmc $set --output-format csv -d $OUT/c$i -- python3 $R/tools/bench_configs.py $CFG > $OUT/c$i.log 2>&1 || { echo "pass $i failed"; tail -3 $OUT/c$i.log; }
  python3 - <<PY
import csv,glob,collections
acc=collections.defaultdict(float)
for f in glob.glob("$OUT/c$i/*/*counter_collection.csv"):
    for r in csv.DictReader(open(f)):
        if "rtiow" in r["Kernel_Name"] and "true>" in r["Kernel_Name"] and float(r["End_Timestamp"])-float(r["Start_Timestamp"]) > 5e7:
            acc[r["Counter_Name"]]+=float(r["Counter_Value"])
for k,v in acc.items(): print(k, v)
PY
done
