#!/bin/bash
# instruction-cache counters of the headline kernel. usage: tools/pmc_icache.sh <tag>
TAG=$1
R=${GRAFT_REPO_ROOT:-/root/repo}
OUT=$R/gpurun_out/$TAG; mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp
i=0
for set in "SQC_ICACHE_REQ SQC_ICACHE_HITS SQC_ICACHE_MISSES" "SQ_IFETCH SQ_WAIT_INST_ANY SQ_BUSY_CYCLES" "SQ_INSTS_VALU SQ_WAVE_CYCLES SQ_WAIT_ANY"; do
  i=$((i+1))
  rocprofv3 --pmc $set --output-format csv -d $OUT/i$i -- python3 $R/bench.py --no-cpu-baseline --steps 1 --warmup 1 --spp 256 > $OUT/i$i.log 2>&1 || { echo "pass $i failed"; tail -3 $OUT/i$i.log; }
  python3 - <<PY
import csv,glob,collections
acc=collections.defaultdict(float)
for f in glob.glob("$OUT/i$i/*/*counter_collection.csv"):
    for r in csv.DictReader(open(f)):
        if "rtiow" in r["Kernel_Name"] and "false>" in r["Kernel_Name"]:
            acc[r["Counter_Name"]]+=float(r["Counter_Value"])
for k,v in acc.items(): print(k, v)
PY
done
