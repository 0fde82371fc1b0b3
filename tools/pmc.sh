#!/bin/bash
# SQ counter passes for the render kernel (separate --pmc passes, at most 8 SQ counters each; kernel-trace / stats never combined
# with --pmc).  usage: tools/pmc.sh <tag> <variant> <spp> [extra bench args]
# Sums every counter over the launches of the TIMED instantiation (name contains "false>") and writes <tag>/summary.txt.
TAG=$1; export RL_RTIOW_KERNEL=$2; SPP=$3; shift 3
R=${GRAFT_REPO_ROOT:-/root/repo}
OUT=$R/gpurun_out/$TAG; mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp
: > $OUT/summary.txt
i=0
for set in "SQ_WAVES SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_ACTIVE_INST_VALU SQ_THREAD_CYCLES_VALU" \
           "SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_WAIT_INST_LDS SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_ACTIVE_INST_LDS SQ_ACTIVE_INST_SCA" \
           "SQ_INSTS_SMEM SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR SQ_ACTIVE_INST_VMEM SQ_INST_CYCLES_SALU SQ_ACTIVE_INST_MISC SQ_INSTS_BRANCH GRBM_GUI_ACTIVE" \
           "SQ_ACTIVE_INST_VALU2 SQ_BUSY_CU_CYCLES SQ_INSTS_VALU_FMA_F64 SQ_INSTS_VALU_ADD_F64 SQ_INSTS_VALU_MUL_F64 SQ_INSTS_VALU_TRANS_F64 SQ_INSTS_VALU_FMA_F32 SQ_INSTS_VALU_ADD_F32" \
           "SQ_INSTS_VALU_MUL_F32 SQ_INSTS_VALU_TRANS_F32 SQ_INSTS_VALU_INT32 SQ_INSTS_VALU_INT64 SQ_INSTS_VALU_CVT SQ_VALU_MFMA_BUSY_CYCLES SQ_INST_LEVEL_LDS SQ_LEVEL_WAVES"; do
  i=$((i+1))
  rocprofv3 --pmc $set --output-format csv -d $OUT/p$i -- python3 $R/bench.py --no-cpu-baseline --no-check --steps 1 --warmup 1 --spp $SPP "$@" > $OUT/p$i.log 2>&1 || { echo "pass $i failed"; tail -3 $OUT/p$i.log; }
  python3 - <<PY | tee -a $OUT/summary.txt
import csv,glob,collections,re
acc=collections.defaultdict(float)
for f in glob.glob("$OUT/p$i/*/*counter_collection.csv"):
    for r in csv.DictReader(open(f)):
        if re.search(r"rtiow_wave_kernel<\d+, \d+, false|rtiow_fast_general_kernel|rtiow_coop_kernel|rtc_kernel|rtc_full_kernel", r["Kernel_Name"]):  # the TIMED (counter-free) instantiations
            acc[r["Counter_Name"]]+=float(r["Counter_Value"])
for k,v in acc.items(): print(k, v)
PY
done
