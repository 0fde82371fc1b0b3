#!/bin/bash
# SQ counter passes for a non-headline config's TIMED kernel (rtiow_fast_general_kernel).  usage: tools/pmc_cfg.sh <tag> <cfg4|cfg5>
# Five separate --pmc passes (never combined with tracing); sums over the timed kernel's launches -> <tag>/summary.txt, and the derived
# per-ray / utilisation figures (same issue-slot model as tools/valu_profile.py) -> <tag>/derived.json
TAG=$1; CFG=$2
R=${GRAFT_REPO_ROOT:-/root/repo}
OUT=$R/gpurun_out/$TAG; mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp
export CFG4_SPP=${CFG4_SPP:-16} CFG5_SPP=${CFG5_SPP:-4}
: > $OUT/summary.txt
i=0
for set in "SQ_WAVES SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_ACTIVE_INST_VALU SQ_THREAD_CYCLES_VALU" \
           "SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_WAIT_INST_LDS SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_ACTIVE_INST_LDS SQ_ACTIVE_INST_SCA" \
           "SQ_INSTS_SMEM SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR SQ_ACTIVE_INST_VMEM SQ_INST_CYCLES_SALU SQ_ACTIVE_INST_MISC SQ_INSTS_BRANCH GRBM_GUI_ACTIVE" \
           "SQ_ACTIVE_INST_VALU2 SQ_BUSY_CU_CYCLES SQ_INSTS_VALU_FMA_F64 SQ_INSTS_VALU_ADD_F64 SQ_INSTS_VALU_MUL_F64 SQ_INSTS_VALU_TRANS_F64 SQ_INSTS_VALU_FMA_F32 SQ_INSTS_VALU_ADD_F32" \
           "SQ_INSTS_VALU_MUL_F32 SQ_INSTS_VALU_TRANS_F32 SQ_INSTS_VALU_INT32 SQ_INSTS_VALU_INT64 SQ_INSTS_VALU_CVT SQ_INST_LEVEL_VMEM SQ_INST_LEVEL_LDS SQ_LEVEL_WAVES"; do
  i=$((i+1))
  rocprofv3 --pmc $set --output-format csv -d $OUT/p$i -- python3 $R/tools/bench_configs.py $CFG > $OUT/p$i.log 2>&1 || { echo "pass $i failed"; tail -3 $OUT/p$i.log; }
  python3 - <<PY | tee -a $OUT/summary.txt
import csv,glob,collections
acc=collections.defaultdict(float)
for f in glob.glob("$OUT/p$i/*/*counter_collection.csv"):
    for r in csv.DictReader(open(f)):
        # every launch of the fast kernel: the timed frame plus bench_configs.py's two 1-spp warm-up frames (accounted for below)
        if "rtiow_fast_general_kernel" in r["Kernel_Name"]:
            acc[r["Counter_Name"]]+=float(r["Counter_Value"])
for k,v in acc.items(): print(k, v)
PY
done
python3 - <<PY | tee $OUT/derived.json
import json
c = {}
for line in open("$OUT/summary.txt"):
    k, v = line.split(); c[k] = float(v)
b = None
for line in open("$OUT/p1.log"):
    if line.startswith("{") and '"rays"' in line: b = json.loads(line)
rays = b["rays"] * (1.0 + 2.0 / b["spp"])  # + two warm-up frames at 1 spp
n64 = c["SQ_INSTS_VALU_ADD_F64"] + c["SQ_INSTS_VALU_MUL_F64"] + c["SQ_INSTS_VALU_FMA_F64"] + c["SQ_INSTS_VALU_INT64"]
ntr = c["SQ_INSTS_VALU_TRANS_F64"]
slots = (c["SQ_INSTS_VALU"] - n64 - ntr) + 2 * n64 + 4 * ntr
cycles = c["GRBM_GUI_ACTIVE"] / 8.0
lanes = c["SQ_THREAD_CYCLES_VALU"] / (64.0 * c["SQ_ACTIVE_INST_VALU"])
print(json.dumps({"config": b["config"], "spp": b["spp"], "rays": rays, "timed_kernel_ms_under_pmc": b["timed_kernel_ms"], "Mrays_s_under_pmc": b["Mrays_s"],
  "note": "counters summed over every launch of rtiow_fast_general_kernel in tools/bench_configs.py (timed frame + two 1-spp warm-up frames; rays scaled accordingly)",
  "valu_wave_insts_per_ray": c["SQ_INSTS_VALU"] / rays, "valu_issue_slots_per_ray": slots / rays, "lanes_active_frac": lanes,
  "valu_issue_frac": slots * 2.0 / (1024.0 * cycles), "rocprof_valubusy_formula": c["SQ_ACTIVE_INST_VALU"] / (256.0 * cycles),
  "vmem_rd_insts_per_ray": c["SQ_INSTS_VMEM_RD"] / rays, "lds_insts_per_ray": c["SQ_INSTS_LDS"] / rays, "salu_insts_per_ray": c["SQ_INSTS_SALU"] / rays,
  "wave_cycle_shares": {"active": c["SQ_ACTIVE_INST_ANY"] / c["SQ_WAVE_CYCLES"], "issue_stalled": c["SQ_WAIT_INST_ANY"] / c["SQ_WAVE_CYCLES"], "waiting": c["SQ_WAIT_ANY"] / c["SQ_WAVE_CYCLES"]},
  "mean_vmem_in_flight_per_wave": c.get("SQ_INST_LEVEL_VMEM", 0.0) / c["SQ_WAVE_CYCLES"] if c.get("SQ_WAVE_CYCLES") else None}))
PY
