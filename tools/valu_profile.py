#!/usr/bin/env python3
"""Turns the SQ counter sums of tools/pmc.sh (gpurun_out/<tag>/summary.txt) into profiles/valu.json: the per-ray instruction
figures of the timed kernel that bench.py's roofline multiplies by ITS OWN rays and kernel time.

usage: tools/valu_profile.py <tag> [hbm_bytes_per_launch]        (run where gpurun_out/<tag>/ is, i.e. in the repo root)

Issue-slot model (/opt/skills/guides/MI355X_MICROARCH.md): a SIMD-32 issues a wave64 binary32 / int32 VALU instruction in 2 cycles =
one SLOT; binary64 arithmetic (v_add / v_mul / v_fma_f64) and 64-bit integer instructions run at half rate = 2 slots (78.6 vs 157.3
TFLOP/s vector peak); binary64 transcendentals (v_rcp / v_rsq / v_sqrt_f64) are counted as 4 slots (quarter rate).  Everything
that SQ_INSTS_VALU counts beyond the typed classes (v_cndmask, v_cmp, v_min / v_max, v_mov, bit operations...) is a 1-slot
instruction."""
import json
import os
import re
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
tag = sys.argv[1]
hbm = float(sys.argv[2]) if len(sys.argv) > 2 else None
d = os.path.join(ROOT, "gpurun_out", tag)
c = {}
for line in open(os.path.join(d, "summary.txt")):
    k, v = line.split()
    c[k] = float(v)
bench = None
for line in open(os.path.join(d, "p1.log")):
    if line.startswith("{"):
        bench = json.loads(line)
rays = bench["config"]["rays_per_step"] * (bench["steps"])  # the timed launches of the pass (the counting step is another kernel)
m = re.search(r"(\d+)x(\d+), (\d+) spp, depth (\d+)", bench["config"]["workload"])
W, H, spp, depth = (int(x) for x in m.groups())
n64 = c["SQ_INSTS_VALU_ADD_F64"] + c["SQ_INSTS_VALU_MUL_F64"] + c["SQ_INSTS_VALU_FMA_F64"] + c["SQ_INSTS_VALU_INT64"]
ntr64 = c["SQ_INSTS_VALU_TRANS_F64"]
n32 = c["SQ_INSTS_VALU"] - n64 - ntr64
slots = n32 + 2.0 * n64 + 4.0 * ntr64
lanes = c["SQ_THREAD_CYCLES_VALU"] / (64.0 * c["SQ_ACTIVE_INST_VALU"])
cycles = c["GRBM_GUI_ACTIVE"] / 8.0  # per-XCD sum / 8 = kernel duration in shader cycles
flop64 = (2.0 * c["SQ_INSTS_VALU_FMA_F64"] + c["SQ_INSTS_VALU_ADD_F64"] + c["SQ_INSTS_VALU_MUL_F64"]) * 64.0 * lanes
typed = ["SQ_INSTS_VALU_ADD_F32", "SQ_INSTS_VALU_MUL_F32", "SQ_INSTS_VALU_FMA_F32", "SQ_INSTS_VALU_TRANS_F32", "SQ_INSTS_VALU_INT32", "SQ_INSTS_VALU_CVT",
         "SQ_INSTS_VALU_ADD_F64", "SQ_INSTS_VALU_MUL_F64", "SQ_INSTS_VALU_FMA_F64", "SQ_INSTS_VALU_TRANS_F64", "SQ_INSTS_VALU_INT64"]
out = {
    "workload_scene": "bouncing_spheres(1)", "width": W, "height": H, "depth": depth, "pmc_spp": spp, "pmc_rays": rays,
    "kernel": "rtiow_wave_kernel<1024,4,false>",
    "valu_wave_insts_per_ray": c["SQ_INSTS_VALU"] / rays,
    "class_wave_insts_per_ray": {k[len("SQ_INSTS_VALU_"):]: c[k] / rays for k in typed} | {"OTHER_32": (c["SQ_INSTS_VALU"] - sum(c[k] for k in typed)) / rays},
    "valu_issue_slots_per_ray": slots / rays,
    "lanes_active_frac": lanes,
    "valu_lane_ops_per_ray_f32_weighted": slots / rays * 64.0 * lanes,
    "valu_issue_frac_measured": slots * 2.0 / (1024.0 * cycles),
    "rocprof_valubusy_formula": c["SQ_ACTIVE_INST_VALU"] / (256.0 * cycles),
    "fp64_flop_per_ray": flop64 / rays,
    "salu_insts_per_ray": c["SQ_INSTS_SALU"] / rays, "lds_insts_per_ray": c["SQ_INSTS_LDS"] / rays,
    "lds_array_cycles_per_ray": c["SQ_LDS_IDX_ACTIVE"] / rays,
    "lds_active_frac_measured": c["SQ_LDS_IDX_ACTIVE"] / (256.0 * cycles),
    "lds_bank_conflict_frac": c["SQ_LDS_BANK_CONFLICT"] / c["SQ_LDS_IDX_ACTIVE"],
    "wave_cycle_shares": {"active": c["SQ_ACTIVE_INST_ANY"] / c["SQ_WAVE_CYCLES"], "issue_stalled": c["SQ_WAIT_INST_ANY"] / c["SQ_WAVE_CYCLES"],
                          "waiting": c["SQ_WAIT_ANY"] / c["SQ_WAVE_CYCLES"]},
    "effective_clock_ghz_under_pmc": cycles / (bench["roofline"]["kernel_ms"] * 1e-3 * bench["steps"]) / 1e9,
    "hbm_bytes_per_launch": hbm,
    "source": f"profiles/r02_{tag}_sq.txt = tools/pmc.sh passes (rocprofv3 --pmc, {W}x{H} {spp} spp, one timed frame = two launches summed); derived by tools/valu_profile.py",
}
print(json.dumps(out, indent=1))
