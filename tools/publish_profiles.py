#!/usr/bin/env python3
"""Copies the counter evidence of one tools/pmc_roofline.py run (gpurun_out/<tag>/<cfg>/) into profiles/:
  profiles/roofline_<cfg>.json    what bench.py's `configs[*].roofline` reads (per-ray figures of the timed kernel)
  profiles/<round>_<cfg>_counters.txt   the raw counter sums (every pass, the timed kernel's launches only)
  profiles/valu.json              (cfg2) what bench.py's headline `roofline` reads
usage: tools/publish_profiles.py <tag> <round, e.g. r03>"""
import json, os, shutil, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
tag, rnd = sys.argv[1], sys.argv[2]
KERNEL = {"cfg2": "rtiow_wave_kernel<1024,4,false>", "cfg3_aa1": "rtc_kernel<256,true>", "cfg3_aa8": "rtc_kernel<256,true>", "cfg4": "rtiow_fast_general_kernel<768,20,false>",
          "cfg5": "rtiow_fast_general_kernel<768,20,false>"}
BOUND = {"cfg2": "valu issue under divergence (scene in LDS; HBM traffic is the framebuffer)",
         "cfg3_aa1": "launch / latency (2.9 M rays per frame: 0.5 ms; scene in LDS; HBM traffic is the framebuffer)",
         "cfg3_aa8": "valu issue (scene in LDS, every lane walks the same ops; HBM traffic is the framebuffer)",
         "cfg4": "latency of dependent fetches served by L2 (2.3 MB scene: 98 % L2 hits) at 3 waves per SIMD",
         "cfg5": "latency of dependent 128-byte node fetches: 80 % L2 hits, the rest from the Infinity Cache / HBM, at 3 waves per SIMD; "
                 "HBM-side traffic is a few percent of peak"}
for cfg in KERNEL:
    d = os.path.join(ROOT, "gpurun_out", tag, cfg)
    f = os.path.join(d, f"roofline_{cfg}.json")
    if not os.path.exists(f):
        continue
    r = json.load(open(f))
    r["kernel"], r["bound"] = KERNEL[cfg], BOUND[cfg]
    r["source"] = f"profiles/{rnd}_{cfg}_counters.txt = " + r["source"]
    shutil.copy(os.path.join(d, "counters.txt"), os.path.join(ROOT, "profiles", f"{rnd}_{cfg}_counters.txt"))
    if cfg == "cfg2":
        w = r["workload"]
        v = {"workload_scene": "bouncing_spheres(1)", "width": w["W"], "height": w["H"], "depth": w["depth"], "pmc_spp": w["spp"], "pmc_rays": r["pmc_rays"]}
        v.update({k: r[k] for k in r if k not in ("workload", "config", "kernel_regex")})
        json.dump(v, open(os.path.join(ROOT, "profiles", "valu.json"), "w"), indent=1)
    else:
        json.dump(r, open(os.path.join(ROOT, "profiles", f"roofline_{cfg}.json"), "w"), indent=1)
    print(cfg, "->", "profiles/valu.json" if cfg == "cfg2" else f"profiles/roofline_{cfg}.json")
