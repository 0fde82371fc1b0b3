#!/bin/bash
# SQ counter passes for a non-headline config. usage: tools/pmc_cfg.sh <tag> <cfg4|cfg5>
TAG=$1; CFG=$2
R=${GRAFT_REPO_ROOT:-/root/repo}
OUT=$R/gpurun_out/$TAG; mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp
export CFG4_SPP=4 CFG5_SPP=1
i=0
for set in "SQ_WAVES SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_ACTIVE_INST_VALU SQ_THREAD_CYCLES_VALU" \
           "SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_WAIT_INST_LDS SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_ACTIVE_INST_LDS SQ_ACTIVE_INST_SCA" \
           "SQ_INSTS_SMEM SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR SQ_ACTIVE_INST_VMEM SQ_INST_CYCLES_SALU SQ_ACTIVE_INST_MISC SQ_INSTS_BRANCH GRBM_GUI_ACTIVE"; do
  i=$((i+1))
  rocprofv3 --pmc $set --output-format csv -d $OUT/p$i -- python3 $R/tools/bench_configs.py $CFG > $OUT/p$i.log 2>&1 || { echo "pass $i failed"; tail -3 $OUT/p$i.log; }
  python3 - <<PY
import csv,glob,collections
acc=collections.defaultdict(float)
for f in glob.glob("$OUT/p$i/*/*counter_collection.csv"):
    for r in csv.DictReader(open(f)):
        if "rtiow" in r["Kernel_Name"] and "true>" in r["Kernel_Name"] and float(r["End_Timestamp"])-float(r["Start_Timestamp"]) > 5e7:
            acc[r["Counter_Name"]]+=float(r["Counter_Value"])
for k,v in acc.items(): print(k, v)
PY
done
