#!/usr/bin/env python3
"""rocprofv3 counter passes over ONE config's timed kernel -> gpurun_out/<out>/roofline_<cfg>.json (copied to profiles/ by hand).

usage (on the GPU box, through gpurun):  python3 tools/pmc_roofline.py <out> <cfg2|cfg3_aa1|cfg3_aa8|cfg4|cfg5> [spp] [sq|mem|ic|all]

Every pass is its own `rocprofv3 --pmc <set> -- python3 tools/cfg_workload.py <cfg> <spp>` (counters only: never combined with a trace), at
most 8 SQ / 4 TCC / 4 TCP counters per pass (FETCH_SIZE takes 3 TCC slots, WRITE_SIZE 2: MI355X_MICROARCH.md "rocprofv3 PMC slots").
Counters are summed over EVERY launch of the timed kernel in the pass and divided by the rays of exactly those launches (the workload prints
them).  Units and corrections per MI355X_MICROARCH.md "HBM": FETCH_SIZE / WRITE_SIZE are KiB on gfx950 and FETCH_SIZE reports half the bytes
fetched, so hbm_bytes = (2 x FETCH_SIZE + WRITE_SIZE) x 1024; Infinity Cache hits are included in FETCH_SIZE (it counts the L2's
memory-side requests), TCC_EA0_RDREQ_DRAM is the part of those that went on to HBM.
Issue-slot model: tools/valu_profile.py."""
import collections
import csv
import glob
import json
import os
import re
import subprocess
import sys

ROOT = os.environ.get("GRAFT_REPO_ROOT") or os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
KERNELS = {"cfg2": r"rtiow_wave_kernel<\d+, \d+, false", "cfg3_aa1": r"rtc_kernel<", "cfg3_aa8": r"rtc_kernel<", "cfg4": r"rtiow_fast_general_kernel",
           "cfg5": r"rtiow_fast_general_kernel"}
SQ = ["SQ_WAVES SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_ACTIVE_INST_VALU SQ_THREAD_CYCLES_VALU",
      "SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_WAIT_INST_LDS SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_ACTIVE_INST_LDS SQ_ACTIVE_INST_SCA",
      "SQ_INSTS_SMEM SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR SQ_ACTIVE_INST_VMEM SQ_INST_CYCLES_SALU SQ_ACTIVE_INST_MISC SQ_INSTS_BRANCH GRBM_GUI_ACTIVE",
      "SQ_ACTIVE_INST_VALU2 SQ_BUSY_CU_CYCLES SQ_INSTS_VALU_FMA_F64 SQ_INSTS_VALU_ADD_F64 SQ_INSTS_VALU_MUL_F64 SQ_INSTS_VALU_TRANS_F64 SQ_INSTS_VALU_FMA_F32 SQ_INSTS_VALU_ADD_F32",
      "SQ_INSTS_VALU_MUL_F32 SQ_INSTS_VALU_TRANS_F32 SQ_INSTS_VALU_INT32 SQ_INSTS_VALU_INT64 SQ_INSTS_VALU_CVT SQ_INST_LEVEL_VMEM SQ_INST_LEVEL_LDS SQ_LEVEL_WAVES"]
MEM = ["FETCH_SIZE GRBM_GUI_ACTIVE",
       "WRITE_SIZE TCC_HIT_sum TCC_MISS_sum",
       "TCC_REQ_sum TCC_READ_sum TCC_EA0_RDREQ_sum TCC_EA0_RDREQ_32B_sum",
       "TCC_EA0_RDREQ_LEVEL_sum TCC_EA0_RDREQ_DRAM_sum TCC_TAG_STALL_sum TCC_BUSY_sum",
       "TCP_TOTAL_CACHE_ACCESSES_sum TCP_TCC_READ_REQ_sum TCP_TCC_READ_REQ_LATENCY_sum TCP_TOTAL_READ_sum",
       "TCP_UTCL1_REQUEST_sum TCP_UTCL1_TRANSLATION_HIT_sum TCP_UTCL1_TRANSLATION_MISS_sum TCP_PENDING_STALL_CYCLES_sum",
       "TCP_TCP_LATENCY_sum TCP_TA_TCP_STATE_READ_sum TCP_TCR_TCP_STALL_CYCLES_sum TCP_GATE_EN1_sum"]
IC = ["SQC_ICACHE_REQ SQC_ICACHE_HITS SQC_ICACHE_MISSES SQC_ICACHE_MISSES_DUPLICATE SQ_IFETCH SQ_WAVE_CYCLES SQ_WAIT_INST_ANY SQ_WAIT_ANY",
      "SQC_TC_INST_REQ SQC_ICACHE_BUSY_CYCLES SQC_DCACHE_REQ SQC_DCACHE_MISSES SQ_IFETCH_LEVEL SQ_INSTS_SMEM SQ_BUSY_CYCLES GRBM_GUI_ACTIVE"]
# what bench.py's headline roofline needs, in six passes: VALU instructions by class, lanes per instruction, LDS, wave-cycle shares, HBM bytes, L2 hit rate
HEADLINE = [SQ[0], SQ[1], SQ[3], SQ[4], MEM[0], MEM[1]]
CONFIG = [SQ[0], SQ[1], SQ[3], SQ[4], MEM[0], MEM[1], MEM[2]]  # bench.py's `configs` entries: the same plus the L2 request count


def measure(cfg, spp="0", which="all", out_dir=None, label=None, timeout_s=None, quiet=False):
    """Runs the counter passes (each its own rocprofv3 child over tools/cfg_workload.py) and returns the derived per-ray figures (dict),
    or None when no pass succeeded.  The calling process must not have touched the GPU."""
    kernel = KERNELS[cfg]
    sets = HEADLINE if which == "headline" else CONFIG if which == "config" else (SQ if which in ("sq", "all") else []) + (MEM if which in ("mem", "all") else []) + (IC if which in ("ic", "all") else [])
    OUT = out_dir
    os.makedirs(OUT, exist_ok=True)
    env = dict(os.environ, TMPDIR="/tmp")
    c = collections.defaultdict(float)
    launches = {}
    work = None
    for i, s in enumerate(sets):
        d = os.path.join(OUT, f"p{i}")
        cmd = ["rocprofv3", "--pmc"] + s.split() + ["--output-format", "csv", "-d", d, "--", "python3", os.path.join(ROOT, "tools", "cfg_workload.py"), cfg, str(spp)]
        try:
            r = subprocess.run(cmd, cwd="/tmp", env=env, stdout=subprocess.PIPE, stderr=subprocess.PIPE, text=True, timeout=timeout_s)
        except (subprocess.TimeoutExpired, OSError) as e:  # (subprocess.run has killed exactly the child it started)
            if not quiet:
                print(f"pass {i} ({s.split()[0]}...): {type(e).__name__}", flush=True)
            continue
        open(os.path.join(OUT, f"p{i}.log"), "w").write(r.stdout + "\n---- stderr ----\n" + r.stderr[-4000:])
        js = [l for l in r.stdout.splitlines() if l.startswith("{")]
        if r.returncode != 0 or not js:
            if not quiet:
                print(f"pass {i} ({s.split()[0]}...) failed rc={r.returncode}", flush=True)
            continue
        w = json.loads(js[-1])
        if work is None:
            work = w
        elif w["rays_all_launches"] != work["rays_all_launches"] and not quiet:
            print(f"pass {i}: rays differ between passes?!", w["rays_all_launches"], work["rays_all_launches"], flush=True)
        seen = collections.defaultdict(float)
        nl = collections.defaultdict(int)
        rows = []
        for f in glob.glob(os.path.join(d, "*", "*counter_collection.csv")):
            for row in csv.DictReader(open(f)):
                if re.search(kernel, row["Kernel_Name"]):
                    seen[row["Counter_Name"]] += float(row["Counter_Value"])
                    nl[row["Counter_Name"]] += 1
                    rows.append('"%s",%s,%s\n' % (row["Kernel_Name"][:80], row["Counter_Name"], row["Counter_Value"]))
        for k, v in seen.items():
            if k == "GRBM_GUI_ACTIVE" and k in c:
                continue
            c[k] = v
            launches[k] = nl[k]
        if not quiet:
            print(f"pass {i}: " + " ".join(f"{k}={v:.6g}" for k, v in seen.items()) + f"  | {w['Mrays_s']:.0f} Mrays/s under pmc", flush=True)
        with open(os.path.join(OUT, f"p{i}_rows.csv"), "w") as fo:  # the raw rows of the timed kernel only (small), for profiles/
            fo.write("Kernel_Name,Counter_Name,Counter_Value\n")
            fo.writelines(rows)
        subprocess.run(["rm", "-rf", d])
    if work is None:
        return None
    rays = float(work["rays_all_launches"])
    with open(os.path.join(OUT, "counters.txt"), "w") as fo:
        for k in sorted(c):
            fo.write(f"{k} {c[k]:.0f}\n")
    res = {"config": cfg, "kernel_regex": kernel, "workload": work, "pmc_rays": rays, "launches_summed": max(launches.values()) if launches else 0}

    def have(*ks):
        return all(k in c for k in ks)

    if have("SQ_INSTS_VALU", "SQ_INSTS_VALU_ADD_F64", "SQ_THREAD_CYCLES_VALU", "GRBM_GUI_ACTIVE"):
        n64 = c["SQ_INSTS_VALU_ADD_F64"] + c["SQ_INSTS_VALU_MUL_F64"] + c["SQ_INSTS_VALU_FMA_F64"] + c["SQ_INSTS_VALU_INT64"]
        ntr = c["SQ_INSTS_VALU_TRANS_F64"]
        slots = (c["SQ_INSTS_VALU"] - n64 - ntr) + 2 * n64 + 4 * ntr
        lanes = c["SQ_THREAD_CYCLES_VALU"] / (64.0 * c["SQ_ACTIVE_INST_VALU"])
        cycles = c["GRBM_GUI_ACTIVE"] / 8.0
        typed = ["SQ_INSTS_VALU_ADD_F32", "SQ_INSTS_VALU_MUL_F32", "SQ_INSTS_VALU_FMA_F32", "SQ_INSTS_VALU_TRANS_F32", "SQ_INSTS_VALU_INT32", "SQ_INSTS_VALU_CVT",
                 "SQ_INSTS_VALU_ADD_F64", "SQ_INSTS_VALU_MUL_F64", "SQ_INSTS_VALU_FMA_F64", "SQ_INSTS_VALU_TRANS_F64", "SQ_INSTS_VALU_INT64"]
        res.update({
            "valu_wave_insts_per_ray": c["SQ_INSTS_VALU"] / rays, "valu_issue_slots_per_ray": slots / rays, "lanes_active_frac": lanes,
            "class_wave_insts_per_ray": {k[len("SQ_INSTS_VALU_"):]: c[k] / rays for k in typed} | {"OTHER_32": (c["SQ_INSTS_VALU"] - sum(c[k] for k in typed)) / rays},
            "valu_lane_ops_per_ray_f32_weighted": slots / rays * 64.0 * lanes,
            "valu_issue_frac_under_pmc": slots * 2.0 / (1024.0 * cycles), "rocprof_valubusy_formula": c["SQ_ACTIVE_INST_VALU"] / (256.0 * cycles),
            "fp64_flop_per_ray": (2.0 * c["SQ_INSTS_VALU_FMA_F64"] + c["SQ_INSTS_VALU_ADD_F64"] + c["SQ_INSTS_VALU_MUL_F64"]) * 64.0 * lanes / rays,
            "salu_insts_per_ray": c["SQ_INSTS_SALU"] / rays, "lds_insts_per_ray": c["SQ_INSTS_LDS"] / rays,
            "vmem_rd_insts_per_ray": c.get("SQ_INSTS_VMEM_RD", 0.0) / rays, "vmem_wr_insts_per_ray": c.get("SQ_INSTS_VMEM_WR", 0.0) / rays,
            "lds_array_cycles_per_ray": c.get("SQ_LDS_IDX_ACTIVE", 0.0) / rays,
            "lds_bank_conflict_frac": (c["SQ_LDS_BANK_CONFLICT"] / c["SQ_LDS_IDX_ACTIVE"]) if c.get("SQ_LDS_IDX_ACTIVE") else None,
            "wave_cycle_shares": {"active": c["SQ_ACTIVE_INST_ANY"] / c["SQ_WAVE_CYCLES"], "issue_stalled": c["SQ_WAIT_INST_ANY"] / c["SQ_WAVE_CYCLES"],
                                  "waiting": c["SQ_WAIT_ANY"] / c["SQ_WAVE_CYCLES"]} if have("SQ_WAIT_ANY", "SQ_WAVE_CYCLES") else None,
            "shader_cycles_under_pmc": cycles})
    if have("FETCH_SIZE"):
        res["fetch_bytes_per_ray"] = 2.0 * c["FETCH_SIZE"] * 1024.0 / rays
    if have("WRITE_SIZE"):
        res["write_bytes_per_ray"] = c["WRITE_SIZE"] * 1024.0 / rays
    if have("FETCH_SIZE", "WRITE_SIZE"):
        res["hbm_bytes_per_ray"] = (2.0 * c["FETCH_SIZE"] + c["WRITE_SIZE"]) * 1024.0 / rays
        res["hbm_bytes_per_launch"] = (2.0 * c["FETCH_SIZE"] + c["WRITE_SIZE"]) * 1024.0 / max(1, launches.get("FETCH_SIZE", 1))
    if have("TCC_HIT_sum", "TCC_MISS_sum"):
        res["l2_hit_rate"] = c["TCC_HIT_sum"] / max(1.0, c["TCC_HIT_sum"] + c["TCC_MISS_sum"])
    if have("TCC_REQ_sum"):
        res["l2_requests_per_ray"] = c["TCC_REQ_sum"] / rays
        res["l2_bytes_per_ray"] = c["TCC_REQ_sum"] * 128.0 / rays  # one request = one 128-byte line (upper bound: partial-line requests count as whole lines)
    if have("TCC_EA0_RDREQ_sum"):
        res["l2_miss_read_requests_per_ray"] = c["TCC_EA0_RDREQ_sum"] / rays
    if have("TCC_EA0_RDREQ_DRAM_sum", "TCC_EA0_RDREQ_sum"):
        res["dram_read_requests_per_ray"] = c["TCC_EA0_RDREQ_DRAM_sum"] / rays
        res["dram_share_of_l2_miss_reads"] = c["TCC_EA0_RDREQ_DRAM_sum"] / max(1.0, c["TCC_EA0_RDREQ_sum"])
    if have("TCC_EA0_RDREQ_LEVEL_sum", "TCC_EA0_RDREQ_sum"):
        res["tcc_ea_read_latency_cycles"] = c["TCC_EA0_RDREQ_LEVEL_sum"] / max(1.0, c["TCC_EA0_RDREQ_sum"])  # mean cycles a read spends beyond L2
    if have("TCP_TCC_READ_REQ_LATENCY_sum", "TCP_TCC_READ_REQ_sum"):
        res["tcp_tcc_read_latency_cycles"] = c["TCP_TCC_READ_REQ_LATENCY_sum"] / max(1.0, c["TCP_TCC_READ_REQ_sum"])  # mean L1-miss round trip
        res["l1_miss_read_requests_per_ray"] = c["TCP_TCC_READ_REQ_sum"] / rays
    if have("TCP_TOTAL_CACHE_ACCESSES_sum", "TCP_TCC_READ_REQ_sum"):
        res["l1_accesses_per_ray"] = c["TCP_TOTAL_CACHE_ACCESSES_sum"] / rays
        res["l1_hit_rate"] = 1.0 - c["TCP_TCC_READ_REQ_sum"] / max(1.0, c["TCP_TOTAL_CACHE_ACCESSES_sum"])
    if have("TCP_UTCL1_REQUEST_sum", "TCP_UTCL1_TRANSLATION_MISS_sum"):
        res["utcl1_miss_rate"] = c["TCP_UTCL1_TRANSLATION_MISS_sum"] / max(1.0, c["TCP_UTCL1_REQUEST_sum"])
    if have("TCP_TCP_LATENCY_sum", "TCP_TOTAL_CACHE_ACCESSES_sum"):
        res["tcp_latency_cycles_per_access"] = c["TCP_TCP_LATENCY_sum"] / max(1.0, c["TCP_TOTAL_CACHE_ACCESSES_sum"])
    if have("SQC_ICACHE_REQ", "SQC_ICACHE_MISSES"):
        res["icache_miss_rate"] = c["SQC_ICACHE_MISSES"] / max(1.0, c["SQC_ICACHE_REQ"])
        res["icache_requests_per_ray"] = c["SQC_ICACHE_REQ"] / rays
        res["icache_misses_per_ray"] = c["SQC_ICACHE_MISSES"] / rays
    res["source"] = (label or f"tools/pmc_roofline.py {cfg} {spp} {which}") + f": rocprofv3 --pmc passes over tools/cfg_workload.py, summed over {res['launches_summed']} launches of the timed kernel"
    json.dump(res, open(os.path.join(OUT, f"roofline_{cfg}.json"), "w"), indent=1)
    return res


if __name__ == "__main__":
    out_tag, cfg = sys.argv[1], sys.argv[2]
    spp = sys.argv[3] if len(sys.argv) > 3 else "0"
    which = sys.argv[4] if len(sys.argv) > 4 else "all"
    r = measure(cfg, spp, which, os.path.join(ROOT, "gpurun_out", out_tag, cfg), label=f"tools/pmc_roofline.py {out_tag} {cfg} {spp} {which}")
    if r is None:
        raise SystemExit("no pass succeeded")
    print(json.dumps(r), flush=True)
