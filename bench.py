#!/usr/bin/env python3
"""bench.py — BASELINE.json's headline metric on MI355X.

Metric: Mrays/sec (primary + secondary) on the RTIOW random-sphere scene
(examples/bouncing_spheres.rs: 488 spheres, 511-node BVH), 1920x1080, 1024 spp, depth 50 (configs[1]).
A "step" = one full render of that frame through the C ABI, scene already resident in HBM, output left in HBM.
With N GPUs the framebuffer is sharded by interleaved rows (row r -> rank r mod N, no data-path collective during
the render) and gathered to rank 0 over RCCL/xGMI once per step — inside the timed region.

  python bench.py                          one GPU
  torch.distributed.run ... bench.py --gpus N   one rank per GPU (the driver's form): rl_rtiow_render_device + dist.gather
  python bench.py --gpus N                 (no launcher) starts exactly that launcher as a child process before touching a GPU
  python bench.py --gpus N --inprocess     ONE process, the library's own multi-GPU entry point (rl_init_multi +
                                           rl_rtiow_render_multi_device: RCCL send/recv inside the library)
It never prints a line whose n_gpus differs from --gpus.

Prints ONE JSON line (rank 0). Extra objects:
  roofline     — the kernel is bound by VALU issue under divergence (the scene lives in LDS / L2; measured HBM traffic is the
                 framebuffer), so the bound is the chip's vector-instruction issue capacity: achieved = VALU lane-operations per
                 ray (rocprofv3 SQ counters of the same kernel on the same workload, profiles/valu.json: instruction counts are a
                 property of the workload) x rays of THIS run / kernel time of THIS run (HIP events on the launch stream),
                 binary64 instructions weighted 2x (they hold the pipe twice as long); peak = 256 CU x 4 SIMD x 32 lanes x 2.4 GHz.
                 frac = valu_issue_frac (share of the SIMDs' issue slots used) x lanes_active_frac (lanes doing work per
                 instruction).  lds_frac = LDS-array busy cycles / CU cycles.
                 `traffic` = measured HBM bytes per launch, algorithmic_bytes = SURVEY.md §8d's figure, both reported, neither the bound.
                 The counter figures are MEASURED IN THE RUN: before this process touches the GPU it starts rocprofv3 --pmc child passes
                 (counters only, tools/pmc_roofline.py) over tools/cfg_workload.py = the same scene / frame / sample count / kernel
                 (`counters_measured_in_this_run`; --no-live-pmc or a missing profiler fall back to profiles/valu.json / roofline_cfg*.json).
  configs      — BASELINE configs[2..4] at their stated sizes after the headline loop (N = 1): each with its own check, roofline figures
                 (counter passes at reduced sample counts, `pmc_spp`) and CPU sample.
  check        — after the timed loop: the timed frame equals the counting kernel's frame bit for bit; 8 full-spp rows of the real
                 frame are re-rendered by the CPU oracle (counters exact, max |err| of the pixel means).
  cpu_baseline — the CPU oracle (C++ restatement of the reference; the Rust reference cannot be built here) timed on this
                 box's host cores over a bounded sample of the same workload.
"""
import argparse
import importlib
import json
import os
import socket
import subprocess
import sys
import time

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)

CLOCK_GHZ, N_CU, N_SIMD, SIMD_LANES = 2.4, 256, 4, 32  # /opt/skills/guides/MI355X_MICROARCH.md (chip-level parameters, SIMD-32)
VALU_PEAK_TLANEOPS = N_CU * N_SIMD * SIMD_LANES * CLOCK_GHZ / 1e3  # 78.6 T lane-ops/s (one binary32 op per lane per clock)


def parse_args():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=2)
    ap.add_argument("--warmup", type=int, default=1)
    ap.add_argument("--width", type=int, default=1920)
    ap.add_argument("--spp", type=int, default=1024)
    ap.add_argument("--depth", type=int, default=50)
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--no-check", action="store_true", help="skip the post-run frame check (profiling passes)")
    ap.add_argument("--check-rows", type=int, default=8)
    ap.add_argument("--no-live-pmc", action="store_true",
                    help="do not collect the headline kernel's rocprofv3 counters in this run (the roofline then uses profiles/valu.json)")
    ap.add_argument("--configs", default="cfg3,cfg4,cfg5", help="non-headline BASELINE configs appended to the line at N = 1 ('' = none)")
    ap.add_argument("--cfg4-spp", type=int, default=512)
    ap.add_argument("--cfg5-spp", type=int, default=256)
    ap.add_argument("--inprocess", action="store_true", help="N GPUs from ONE process through rl_init_multi / rl_rtiow_render_multi_device")
    ap.add_argument("--emulate-shard", type=int, default=0,
                    help="single-GPU rehearsal of an N-way shard: render only rows 0 mod N (not the headline metric)")
    return ap.parse_args()


def spawn_ranks(args):
    """--gpus N without a launcher: start N fresh ranks under torch.distributed.run BEFORE this process touches a GPU."""
    import torch
    have = torch.cuda.device_count()  # counting devices does not initialise the GPU
    if have < args.gpus and os.environ.get("RL_BENCH_REHEARSAL") != "1":  # (rehearsal: every rank on GPU 0, labelled NOT a measurement)
        raise SystemExit(f"bench.py --gpus {args.gpus}: only {have} GPU(s) visible; refusing to print a mislabelled line")
    with socket.socket() as s:
        s.bind(("127.0.0.1", 0))
        port = s.getsockname()[1]
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", f"--nproc-per-node={args.gpus}", "--master-addr", "127.0.0.1",
           "--master-port", str(port), os.path.abspath(__file__)] + sys.argv[1:]
    raise SystemExit(subprocess.run(cmd).returncode)


def live_valu_profile(args):
    """The headline kernel's per-ray counter figures MEASURED IN THIS RUN: six rocprofv3 --pmc child passes (counters only) over
    tools/cfg_workload.py cfg2 — the same scene, frame, sample count and kernel the timed loop below renders — started BEFORE this process
    touches the GPU (a process that holds the GPU must not spawn programs on this pool).  None when rocprofv3 is missing or a pass fails:
    the roofline then falls back to profiles/valu.json and says so."""
    import shutil
    import tempfile
    if args.no_live_pmc or args.gpus != 1 or args.inprocess or args.emulate_shard > 1 or os.environ.get("WORLD_SIZE") is not None:
        return None
    if args.width != 1920 or args.depth != 50 or shutil.which("rocprofv3") is None:
        return None
    # under a profiler already (rocprofv3 -- python3 bench.py): its preloaded library has initialised the GPU in this very process
    if "rocprof" in os.environ.get("LD_PRELOAD", "").lower() or any(k.startswith(("ROCPROF", "ROCP_")) for k in os.environ):
        return None
    try:
        sys.path.insert(0, os.path.join(ROOT, "tools"))
        pmc = importlib.import_module("pmc_roofline")
        t0 = time.perf_counter()
        out_dir = tempfile.mkdtemp(prefix="rl_bench_pmc_")
        r = pmc.measure("cfg2", str(args.spp), "headline", out_dir=out_dir, label="bench.py", timeout_s=150, quiet=True)
        shutil.rmtree(out_dir, ignore_errors=True)
        need = ("valu_lane_ops_per_ray_f32_weighted", "valu_issue_slots_per_ray", "lanes_active_frac", "lds_array_cycles_per_ray", "valu_wave_insts_per_ray")
        if r is None or any(r.get(k) is None for k in need):
            return None
        w = r["workload"]
        r.update({"workload_scene": "bouncing_spheres(1)", "width": w["W"], "height": w["H"], "depth": w["depth"], "pmc_spp": w["spp"], "kernel": "rtiow_wave_kernel<1024,4,false>",
                  "live": True, "live_seconds": time.perf_counter() - t0,
                  "source": f"MEASURED IN THIS RUN: six rocprofv3 --pmc child passes over tools/cfg_workload.py cfg2 {w['spp']} (tools/pmc_roofline.py `headline` set: "
                            f"SQ instruction classes, lanes per instruction, LDS, wave-cycle shares, FETCH_SIZE, WRITE_SIZE + L2 hits) before the timed loop, "
                            f"{time.perf_counter() - t0:.0f} s"})
        return r
    except Exception as e:  # noqa: BLE001 — the bench line must not depend on the profiler
        print(f"[bench] live counter passes failed ({type(e).__name__}: {e}); using profiles/valu.json", file=sys.stderr, flush=True)
        return None


LIVE_CFG = {}  # tag -> per-ray counter figures of a `configs` entry's timed kernel measured in this run (live_cfg_profiles)


def live_cfg_profiles(args):
    """The same for the other BASELINE configs' timed kernels (rtc_kernel, rtiow_fast_general_kernel on cfg 4 / cfg 5), at reduced sample
    counts (per-ray instruction and byte counts hardly depend on them; `pmc_spp` says which) so that the passes add about 1.5 minutes."""
    import shutil
    import tempfile
    if args.no_live_pmc or not args.configs or shutil.which("rocprofv3") is None or args.gpus != 1 or args.inprocess or args.emulate_shard > 1:
        return
    if os.environ.get("WORLD_SIZE") is not None or "rocprof" in os.environ.get("LD_PRELOAD", "").lower() or any(k.startswith(("ROCPROF", "ROCP_")) for k in os.environ):
        return
    which = [c.strip() for c in args.configs.split(",") if c.strip()]
    jobs = ([("cfg3_aa1", "20"), ("cfg3_aa8", "2")] if "cfg3" in which else []) + ([("cfg4", str(min(args.cfg4_spp, 128)))] if "cfg4" in which else []) + \
           ([("cfg5", str(min(args.cfg5_spp, 64)))] if "cfg5" in which else [])
    try:
        sys.path.insert(0, os.path.join(ROOT, "tools"))
        pmc = importlib.import_module("pmc_roofline")
        for tag, spp in jobs:
            t0 = time.perf_counter()
            out_dir = tempfile.mkdtemp(prefix="rl_bench_pmc_")
            r = pmc.measure(tag, spp, "config", out_dir=out_dir, label="bench.py", timeout_s=150, quiet=True)
            shutil.rmtree(out_dir, ignore_errors=True)
            if r is None or r.get("valu_lane_ops_per_ray_f32_weighted") is None:
                continue
            base = cfg_profile(tag) or {}
            r.update({"kernel": base.get("kernel"), "bound": base.get("bound"), "live": True, "pmc_spp": spp,
                      "source": f"MEASURED IN THIS RUN: rocprofv3 --pmc child passes over tools/cfg_workload.py {tag} {spp} before the timed loop, {time.perf_counter() - t0:.0f} s"})
            LIVE_CFG[tag] = r
            print(f"[bench] live counters {tag}: {time.perf_counter() - t0:.0f} s", file=sys.stderr, flush=True)
    except Exception as e:  # noqa: BLE001
        print(f"[bench] live counter passes for the configs failed ({type(e).__name__}: {e}); using profiles/", file=sys.stderr, flush=True)


def valu_profile(W, H, spp, depth, live=None):
    """SQ-counter figures of the timed kernel on this workload: measured in this run (live_valu_profile) or, failing that, profiles/valu.json."""
    if live is not None and live.get("width") == W and live.get("depth") == depth and live.get("pmc_spp") == spp:
        return live
    path = os.path.join(ROOT, "profiles", "valu.json")
    try:
        vj = json.load(open(path))
    except Exception:
        return None
    return vj if vj.get("workload_scene") == "bouncing_spheres(1)" and vj.get("depth") == depth and vj.get("width") == W else None


def cfg_profile(tag):
    """rocprofv3 figures of a non-headline config's timed kernel (profiles/roofline_<tag>.json, written by tools/pmc_mem.sh + tools/pmc_cfg.sh)."""
    try:
        return json.load(open(os.path.join(ROOT, "profiles", f"roofline_{tag}.json")))
    except Exception:
        return None


def cfg_roofline(tag, rays, kernel_s, alg_bytes):
    """The roofline object of one `configs` entry: this run's rays and kernel time x the per-ray counter figures of profiles/."""
    roof = {"algorithmic_bytes_per_ray": alg_bytes / max(1.0, rays), "algorithmic_gbs": alg_bytes / kernel_s / 1e9,
            "hbm_peak_gbs": 8000.0, "valu_peak_tlaneops": VALU_PEAK_TLANEOPS}
    pj = LIVE_CFG.get(tag) or cfg_profile(tag)
    if not pj:
        roof["note"] = f"no profiles/roofline_{tag}.json: counter figures not available"
        return roof
    rps = rays / kernel_s
    roof["kernel"] = pj.get("kernel")
    roof["per_ray_figures_from"] = pj.get("source")
    roof["counters_measured_in_this_run"] = bool(pj.get("live"))
    if pj.get("pmc_spp") is not None:
        roof["pmc_spp"] = pj.get("pmc_spp")
    if pj.get("valu_lane_ops_per_ray_f32_weighted") is not None:
        ach = pj["valu_lane_ops_per_ray_f32_weighted"] * rps / 1e12
        roof.update({"valu_achieved_tlaneops": ach, "valu_frac": ach / VALU_PEAK_TLANEOPS,
                     "valu_issue_frac": pj["valu_issue_slots_per_ray"] * rps / (N_CU * N_SIMD * CLOCK_GHZ * 1e9 / 2.0),
                     "lanes_active_frac": pj.get("lanes_active_frac"), "wave_cycle_shares": pj.get("wave_cycle_shares")})
    if pj.get("hbm_bytes_per_ray") is not None:  # (2 x FETCH_SIZE + WRITE_SIZE) x 1024 / rays: bytes that left L2 for the Infinity Cache / HBM
        roof.update({"hbm_bytes_per_ray": pj["hbm_bytes_per_ray"], "hbm_gbs": pj["hbm_bytes_per_ray"] * rps / 1e9,
                     "hbm_frac": pj["hbm_bytes_per_ray"] * rps / 8e12})
    for k in ("l2_hit_rate", "l2_bytes_per_ray", "dram_read_bytes_per_ray", "l1_hit_rate", "tcp_tcc_read_latency_cycles", "tcc_ea_read_latency_cycles",
              "utcl1_miss_rate", "bound"):
        if pj.get(k) is not None:
            roof[k] = pj[k]
    if pj.get("l2_bytes_per_ray") is not None:
        roof["l2_gbs"] = pj["l2_bytes_per_ray"] * rps / 1e9
    if pj.get("dram_read_bytes_per_ray") is not None:
        roof["dram_read_gbs"] = pj["dram_read_bytes_per_ray"] * rps / 1e9
    return roof


def cpu_sample_rtiow(rl, oracle, np, world, p, threads, target_s, stride):
    """The CPU oracle over a bounded pixel sample of an RTIOW workload (every `stride`-th row and column, spp calibrated by a 1-spp probe)."""
    cam1 = rl.Camera(rl.CameraParams(**{**p.__dict__, "samples_per_pixel": 1}))
    W, H = cam1.c.image_width, cam1.c.image_height
    gx, gy = np.meshgrid(np.arange(0, W, stride, dtype=np.uint32), np.arange(0, H, stride, dtype=np.uint32))
    gx, gy = gx.ravel(), gy.ravel()
    c0 = time.perf_counter()
    oracle.rtiow_render_pixels(world.desc, cam1.c, gx, gy, threads=threads)
    pdt = max(time.perf_counter() - c0, 1e-3)
    cspp = int(max(1, min(p.samples_per_pixel, target_s / pdt)))
    ccam = rl.Camera(rl.CameraParams(**{**p.__dict__, "samples_per_pixel": cspp}))
    cst = {}
    c0 = time.perf_counter()
    oracle.rtiow_render_pixels(world.desc, ccam.c, gx, gy, threads=threads, stats=cst)
    cdt = time.perf_counter() - c0
    return {"value": cst["rays"] / cdt / 1e6, "unit": "Mrays/s", "cores": threads, "kind": "port",
            "sample": f"same scene/camera {W}x{H} depth {p.max_depth}: every {stride}th row and column ({len(gx)} pixels), {cspp} spp, {cst['rays']} rays in {cdt:.1f} s "
                      f"on {threads} threads (one task per pixel); CPU restatement of the reference algorithm (oracle/), not the Rust reference"}


def bench_rtiow_config(rl, oracle, np, torch, dev, name, tag, baseline_config, world, p, threads, check_rows, cpu_target_s, cpu_stride, log):
    """One non-headline RTIOW config at its stated size: counting render (the reference's counters), ONE timed counter-free frame (HIP events on
    the launch stream), the timed frame against the counting frame and against oracle rows, roofline figures, CPU sample."""
    cam = rl.Camera(p)
    W, H = cam.c.image_width, cam.c.image_height
    stream = torch.cuda.current_stream(dev)
    warm = rl.Camera(rl.CameraParams(**{**p.__dict__, "samples_per_pixel": 1}))
    buf = torch.zeros((H, W, 3), dtype=torch.float64, device=dev)
    warm.render_device(world, buf.data_ptr(), stream=stream.cuda_stream)  # code-object load, clocks
    rl.api.render_status(world)
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record(stream)
    cam.render_device(world, buf.data_ptr(), stream=stream.cuda_stream)
    e1.record(stream)
    status = rl.api.render_status(world)
    torch.cuda.synchronize(dev)
    timed_ms = e0.elapsed_time(e1)
    log(f"{tag}: timed frame {timed_ms:.0f} ms, {status['rays']} rays")
    st = {}
    cbuf = torch.zeros((H, W, 3), dtype=torch.float64, device=dev)
    cam.render_device(world, cbuf.data_ptr(), stream=stream.cuda_stream, stats=st)  # counting (reference-order) kernel, untimed
    torch.cuda.synchronize(dev)
    log(f"{tag}: counting frame {st['kernel_ms']:.0f} ms")
    alg = 64.0 * st["node_tests"] + 64.0 * st["sphere_tests"] + 128.0 * st["planar_tests"] + 216.0 * st["instance_enters"] + 208.0 * st["rays"]
    chk = {"timed_frame_equals_counting_frame": bool(torch.equal(buf, cbuf)), "timed_rays_equal_counting_rays": int(status["rays"]) == int(st["rays"]),
           "flagged": int(st["flagged"]), "slow_traces": int(status.get("slow_traces", 0))}
    del cbuf
    rows = max(1, min(check_rows, H))
    cstep = max(1, H // rows)
    cfirst = cstep // 2
    ys = np.arange(cfirst, H, cstep, dtype=np.uint32)
    rows_buf = torch.zeros((len(ys), W, 3), dtype=torch.float64, device=dev)
    rs = {}
    cam.render_device(world, rows_buf.data_ptr(), stream=stream.cuda_stream, row_first=cfirst, row_step=cstep, stats=rs)
    torch.cuda.synchronize(dev)
    gx, gy = np.meshgrid(np.arange(W, dtype=np.uint32), ys)
    cs = {}
    c0 = time.perf_counter()
    cpu = oracle.rtiow_render_pixels(world.desc, cam.c, gx.ravel(), gy.ravel(), stats=cs).reshape(len(ys), W, 3)
    cdt = time.perf_counter() - c0
    timed_rows = buf[torch.as_tensor(ys.astype(np.int64), device=dev)]
    chk.update({"rows_checked": int(len(ys)), "pixels_checked": int(len(ys) * W), "spp": p.samples_per_pixel,
                "timed_rows_equal_row_shard_render": bool(torch.equal(timed_rows, rows_buf)),
                "counters_equal": all(int(rs[k]) == int(cs[k]) for k in ("rays", "node_tests", "sphere_tests", "planar_tests", "instance_enters", "rng_words", "flagged")),
                "max_abs_err": float(np.abs(timed_rows.cpu().numpy() - cpu).max() / max(1, p.samples_per_pixel)), "tolerance": 1e-4, "oracle_seconds": cdt})
    log(f"{tag}: oracle rows {cdt:.1f} s")
    rays = float(status["rays"])
    out = {"baseline_config": baseline_config, "workload": name, "rays_per_step": rays, "steps": 1, "ms_per_step": timed_ms, "Mrays_s": rays / timed_ms / 1e3,
           "counting_kernel_ms": st["kernel_ms"], "counting_Mrays_s": st["rays"] / st["kernel_ms"] / 1e3,
           "per_ray_reference_counts": {k: st[k] / max(1, st["rays"]) for k in ("node_tests", "sphere_tests", "planar_tests", "instance_enters")},
           "check": chk, "roofline": cfg_roofline(tag, rays, timed_ms * 1e-3, alg)}
    out["cpu_baseline"] = cpu_sample_rtiow(rl, oracle, np, world, p, threads, cpu_target_s, cpu_stride)
    log(f"{tag}: cpu sample done")
    return out


def bench_rtc_config(rl, oracle, np, torch, dev, world, aa, frames, threads, tag, log):
    """BASELINE configs[2] (RTC teapot, 1920x1080): `frames` back-to-back frames through rl_rtc_render_device, HIP events on the launch stream."""
    cam = world.camera
    W, H = cam.hsize, cam.vsize
    stream = torch.cuda.current_stream(dev)
    buf = torch.zeros((H, W, 3), dtype=torch.float64, device=dev)
    st = {}
    cbuf = torch.zeros((H, W, 3), dtype=torch.float64, device=dev)
    world.render_device(cbuf.data_ptr(), aa, stream=stream.cuda_stream, stats=st)  # counting instantiation (also the warm-up)
    for _ in range(max(1, min(frames, 200))):  # untimed: the GPU has idled through the CPU legs above — bring its clocks back before a 50 ms loop
        world.render_device(buf.data_ptr(), aa, stream=stream.cuda_stream)
    rl.api.render_status(world)
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record(stream)
    for _ in range(frames):
        world.render_device(buf.data_ptr(), aa, stream=stream.cuda_stream)
    e1.record(stream)
    status = rl.api.render_status(world)
    torch.cuda.synchronize(dev)
    ms = e0.elapsed_time(e1) / frames
    rays = float(st["rays"])
    cstep = 40
    cs = {}
    c0 = time.perf_counter()
    cpu = oracle.rtc_render(world.desc, cam, aa=aa, row_first=0, row_step=cstep, threads=threads, stats=cs)
    cdt = time.perf_counter() - c0
    rs = {}
    world.render(aa, row_first=0, row_step=cstep, stats=rs)
    chk = {"timed_frame_equals_counting_frame": bool(torch.equal(buf, cbuf)), "timed_rays_equal_counting_rays": int(status["rays"]) == int(st["rays"]),
           "rows_checked": int(cpu.shape[0]), "counters_equal": all(int(rs[k]) == int(cs[k]) for k in ("rays", "node_tests", "planar_tests")),
           "max_abs_err": float(np.abs(buf.cpu().numpy()[0::cstep] - cpu).max()), "tolerance": 1e-4}
    alg = 48.0 * st["node_tests"] + 72.0 * st["planar_tests"]
    out = {"baseline_config": "configs[2]", "workload": f"RTC teapot-low.obj (240 triangles, 1 light), {W}x{H}, Phong, AA {aa} ({aa * aa} rays per pixel + shadow rays)",
           "rays_per_step": rays, "steps": frames, "ms_per_step": ms, "Mrays_s": rays / ms / 1e3,
           "per_ray_reference_counts": {"bounds_tests": st["node_tests"] / rays, "triangle_tests": st["planar_tests"] / rays},
           "check": chk, "roofline": cfg_roofline(tag, rays, ms * 1e-3, alg),
           "cpu_baseline": {"value": cs["rays"] / cdt / 1e6, "unit": "Mrays/s", "cores": threads, "kind": "port",
                            "sample": f"same scene/camera, AA {aa}, rows y%{cstep}==0 ({cpu.shape[0]} rows), {cs['rays']} rays in {cdt:.2f} s on {threads} threads; "
                                      "CPU restatement of the reference algorithm (oracle/), not the Rust reference"}}
    log(f"{tag}: {ms:.3f} ms per frame")
    return out


def other_configs(args, rl, np, torch, dev):
    """BASELINE configs[2..4] on this GPU, after the headline loop: bounded (about two minutes), each entry with its own check, roofline figures and
    CPU sample.  configs[3] at its stated 512 spp; configs[4] at --cfg5-spp (256; its stated 4096 spp take two minutes per frame)."""
    import gzip
    sys.path.insert(0, os.path.join(ROOT, "oracle"))
    oracle = importlib.import_module("rl_oracle")
    try:
        threads = min(oracle.hardware_threads(), len(os.sched_getaffinity(0)))
    except AttributeError:
        threads = oracle.hardware_threads()
    t_begin = time.perf_counter()

    def log(msg):  # progress on stderr: a silent multi-minute run looks hung to the launcher
        print(f"[bench configs {time.perf_counter() - t_begin:6.1f} s] {msg}", file=sys.stderr, flush=True)

    G = os.path.join(ROOT, "tests", "golden")
    which = [c.strip() for c in args.configs.split(",") if c.strip()]
    res = []
    if "cfg3" in which:
        w = rl.RtcWorld.test_obj_scene(open(os.path.join(G, "teapot-low.obj"), "rb").read(), 1920, 1080)
        res.append(bench_rtc_config(rl, oracle, np, torch, dev, w, 1, 200, threads, "cfg3_aa1", log))
        res.append(bench_rtc_config(rl, oracle, np, torch, dev, w, 8, 5, threads, "cfg3_aa8", log))
        del w
    if "cfg4" in which or "cfg5" in which:
        from PIL import Image
        tex = np.asarray(Image.open(os.path.join(G, "spot_texture.png")).convert("RGB"))
        obj = gzip.open(os.path.join(G, "spot_triangulated.obj.gz"), "rb").read()
    if "cfg4" in which:
        w = rl.World.cow_scene(obj, tex)
        p = w.params
        p.aspect_ratio, p.image_width, p.samples_per_pixel = 16.0 / 9.0, 3840, args.cfg4_spp
        res.append(bench_rtiow_config(rl, oracle, np, torch, dev,
                                      f"RTIOW examples/cow.rs scene (5856 textured triangles under scale/rotate_y/translate + 6 quads), 3840x2160, {p.samples_per_pixel} spp, depth {p.max_depth}, on ONE GPU",
                                      "cfg4", "configs[3]", w, p, threads, 8, 8.0, 16, log))
        del w
    if "cfg5" in which:
        c0 = time.perf_counter()
        w = rl.World.stress_scene(1000, 2, obj, tex, device_bvh=True)
        build_s = time.perf_counter() - c0
        log(f"cfg5: scene built in {build_s:.1f} s")
        p = w.params
        p.samples_per_pixel = args.cfg5_spp
        e = bench_rtiow_config(rl, oracle, np, torch, dev,
                               f"1,000,000 random spheres + 93,696-triangle mesh (DESIGN.md section 8), 3840x2160, {p.samples_per_pixel} spp (BASELINE: 4096), depth {p.max_depth}, on ONE GPU",
                               "cfg5", "configs[4]", w, p, threads, 8, 8.0, 48, log)
        e["scene_build_s"] = build_s
        res.append(e)
        del w
    log("done")
    return res


def main():
    args = parse_args()
    ws_env = os.environ.get("WORLD_SIZE")
    if ws_env is None and args.gpus > 1 and not args.inprocess:
        spawn_ranks(args)
    world_size = int(ws_env) if ws_env is not None else 1
    live = live_valu_profile(args)  # (child processes: before anything here touches the GPU)
    if world_size == 1:
        live_cfg_profiles(args)
    if args.inprocess:
        if world_size != 1:
            raise SystemExit("--inprocess is a single-process mode: do not start it under a launcher")
    elif world_size != args.gpus:
        raise SystemExit(f"--gpus {args.gpus} but WORLD_SIZE={world_size}")

    import numpy as np
    import torch
    import torch.distributed as dist

    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    if not torch.cuda.is_available():
        raise SystemExit("bench.py needs a GPU: the HIP path has no CPU fallback")
    # RL_BENCH_REHEARSAL=1: every rank uses GPU 0 and the gather goes over gloo through host memory — a way to run the
    # N-rank code path on a one-GPU box (not a measurement; the JSON line says so)
    rehearsal = os.environ.get("RL_BENCH_REHEARSAL") == "1" and world_size > 1
    if rehearsal:
        local_rank = 0
    torch.cuda.set_device(local_rank)
    dev = torch.device("cuda", local_rank)
    if world_size > 1:
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        if rehearsal:
            dist.init_process_group("gloo", rank=rank, world_size=world_size)
        else:
            dist.init_process_group("nccl", rank=rank, world_size=world_size, device_id=dev)

    # the collective layer describes itself: which backend, how many ranks it really spans (an all-reduce of ones)
    comm = None
    if world_size > 1:
        ones = torch.ones(1, dtype=torch.float64, device="cpu" if rehearsal else dev)
        dist.all_reduce(ones, op=dist.ReduceOp.SUM)
        comm = {"backend": dist.get_backend() + (" (REHEARSAL on one GPU through host memory: NOT RCCL)" if rehearsal else " (= RCCL on ROCm)"),
                "world_size": dist.get_world_size(), "allreduce_ones": int(ones.item())}

    rl = importlib.import_module("rendering-learning_amd")
    sharding = importlib.import_module("rendering-learning_amd.sharding")
    G = args.gpus if args.inprocess else world_size
    if args.inprocess:
        emu = int(os.environ.get("RL_BENCH_EMULATE_DEVICES", "0"))  # one-GPU box: G contexts on GPU 0 (NOT a measurement)
        got = rl.api.init_multi(0 if emu else args.gpus, emulate=emu and args.gpus)
        if got != args.gpus:
            raise SystemExit(f"--gpus {args.gpus} --inprocess: the library drives {got} device(s)")
    else:
        rl.init(local_rank)

    # ---- workload: configs[1] of BASELINE.json
    world = rl.World.bouncing_spheres(1)
    p = world.params
    p.image_width, p.samples_per_pixel, p.max_depth = args.width, args.spp, args.depth
    cam = rl.Camera(p)
    W, H = cam.c.image_width, cam.c.image_height
    row_first, row_step = (0, 1) if args.inprocess else (rank, G)
    if args.emulate_shard > 1:
        row_first, row_step = 0, args.emulate_shard
    max_rows = rl.api.rows_for(H, 0, row_step)
    shard = torch.zeros((max_rows, W, 3), dtype=torch.float64, device=dev)
    gathered = [torch.zeros_like(shard) for _ in range(G)] if (world_size > 1 and rank == 0) else None
    frame = torch.zeros((H, W, 3), dtype=torch.float64, device=dev) if (rank == 0 and world_size > 1) else None
    stream = torch.cuda.current_stream(dev)

    def exchange():  # the one exchange step of the path: framebuffer rows -> rank 0 over RCCL/xGMI
        if rehearsal:
            full = sharding.gather_frame(shard.cpu(), H, rank, G)
            if rank == 0:
                frame.copy_(full)
        else:
            sharding.gather_frame(shard, H, rank, G, frame=frame, gathered=gathered)

    def step(stats=None):
        if args.inprocess:  # the library renders on its own streams and gathers itself; rl_render_status waits
            cam.render_multi_device(world, shard.data_ptr(), stats=stats)
            return
        cam.render_device(world, shard.data_ptr(), stream=stream.cuda_stream, row_first=row_first, row_step=row_step, stats=stats)
        if world_size > 1:
            exchange()

    def finish():
        torch.cuda.synchronize(dev)
        if args.inprocess:
            return rl.api.render_status(world)
        return None

    # ---- counters for this exact workload (deterministic: identical for every launch) — untimed; keep the counting frame
    st = {}
    step(stats=st)
    torch.cuda.synchronize(dev)
    counting_frame = (frame if world_size > 1 else shard).clone() if rank == 0 else None
    for _ in range(max(0, args.warmup - 1)):
        step()
    finish()
    if world_size > 1:  # the exchange step alone (untimed run): rows already rendered, one gather to rank 0
        dist.barrier()
        torch.cuda.synchronize(dev)
        g0 = time.perf_counter()
        exchange()
        torch.cuda.synchronize(dev)
        comm["gather_ms"] = (time.perf_counter() - g0) * 1e3
        comm["gather_bytes_per_peer"] = int(shard.numel() * 8)
    if args.inprocess:
        L = rl.api.render_lib()
        comm = {"mode": "one process, rl_init_multi", "devices": int(L.rl_device_count()), "library_uses_rccl": bool(L.rl_debug_multi_uses_rccl()),
                "librccl_loadable": bool(L.rl_debug_rccl_loadable()),
                "exchange": "ncclSend / ncclRecv in one group (rl_multi.hip)" if L.rl_debug_multi_uses_rccl() else "hipMemcpyPeerAsync (peer-copy fallback or emulated contexts)"}
    if world_size > 1:
        dist.barrier()
    torch.cuda.synchronize(dev)

    evs = [(torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)) for _ in range(args.steps)]
    t0 = time.perf_counter()
    for k in range(args.steps):
        if args.inprocess:
            step()
            continue
        evs[k][0].record(stream)
        cam.render_device(world, shard.data_ptr(), stream=stream.cuda_stream, row_first=row_first, row_step=row_step)
        evs[k][1].record(stream)
        if world_size > 1:
            exchange()
    fin = finish()
    if world_size > 1:
        dist.barrier()
    torch.cuda.synchronize(dev)
    elapsed = time.perf_counter() - t0
    if args.inprocess:
        kernel_ms = elapsed / max(1, args.steps) * 1e3  # wall per step (the library's streams are not torch's)
        timed_status = fin
    else:
        kernel_ms = sum(a.elapsed_time(b) for a, b in evs) / max(1, args.steps)
        timed_status = rl.api.render_status(world)  # rays / flagged counted by the TIMED kernel itself (last step)

    tot = torch.tensor([float(st["rays"]), float(st["node_tests"]), float(st["sphere_tests"]), elapsed, kernel_ms, float(timed_status["rays"])],
                       dtype=torch.float64, device="cpu" if rehearsal else dev)
    if world_size > 1:
        mx = tot.clone()
        dist.all_reduce(tot, op=dist.ReduceOp.SUM)
        dist.all_reduce(mx, op=dist.ReduceOp.MAX)
        elapsed = float(mx[3])
        kernel_ms_max = float(mx[4])
    else:
        kernel_ms_max = kernel_ms
    rays, nodes, spheres, timed_rays = float(tot[0]), float(tot[1]), float(tot[2]), float(tot[5])
    if world_size > 1:  # every rank's own kernel time (HIP events on its launch stream)
        per_rank = [torch.zeros(1, dtype=torch.float64, device="cpu" if rehearsal else dev) for _ in range(world_size)]
        dist.all_gather(per_rank, torch.tensor([kernel_ms], dtype=torch.float64, device="cpu" if rehearsal else dev))
        comm["per_rank_kernel_ms"] = [float(t.item()) for t in per_rank]

    if rank == 0:
        ms_per_step = elapsed / args.steps * 1e3
        value = rays * args.steps / elapsed / 1e6
        rank_rays = float(st["rays"]) if not args.inprocess else rays / G  # rank 0's kernel (per-GPU share in-process)
        k_s = kernel_ms * 1e-3
        launches = 2 if (args.spp >= 64 and os.environ.get("RL_LPT", "1") != "0") else 1
        # SURVEY.md §8d's algorithmic bytes (what a scene-from-HBM traversal would move): reported, NOT the bound
        alg_bytes = 64.0 * st["node_tests"] + 64.0 * st["sphere_tests"] + 208.0 * st["rays"]
        vp = valu_profile(W, H, args.spp, args.depth, live)
        roof = {"bound": "valu", "unit": "Tlane-op/s", "peak": VALU_PEAK_TLANEOPS, "kernel": vp["kernel"] if vp else "rtiow_wave_kernel<1024,4,false>",
                "kernel_ms": kernel_ms, "kernel_ms_max_rank": kernel_ms_max, "launches_per_step": launches, "kernel_avg_launch_ms": kernel_ms / launches,
                "algorithmic_bytes_per_launch": alg_bytes, "algorithmic_gbs": alg_bytes / k_s / 1e9, "traffic": None}
        if vp:
            lane_ops = vp["valu_lane_ops_per_ray_f32_weighted"]   # lanes x instructions, binary64 instructions counted twice
            wave_insts = vp["valu_issue_slots_per_ray"]            # wave-instructions, binary64 counted twice (2-cycle slots)
            achieved = lane_ops * rank_rays / k_s / 1e12
            issue = wave_insts * rank_rays / k_s / (N_CU * N_SIMD * CLOCK_GHZ * 1e9 / 2.0)
            lds_frac = vp["lds_array_cycles_per_ray"] * rank_rays / k_s / (N_CU * CLOCK_GHZ * 1e9)  # LDS-array busy cycles / CU cycles
            roof.update({"achieved": achieved, "frac": achieved / VALU_PEAK_TLANEOPS,
                         "valu_issue_frac": issue, "lanes_active_frac": vp["lanes_active_frac"],
                         "lds_frac": lds_frac, "lds_bank_conflict_frac": vp["lds_bank_conflict_frac"],
                         "valu_wave_insts_per_ray": vp["valu_wave_insts_per_ray"], "valu_issue_slots_per_ray": wave_insts,
                         "fp64_flop_per_ray": vp.get("fp64_flop_per_ray"),
                         "fp64_tflops": (vp["fp64_flop_per_ray"] * rank_rays / k_s / 1e12) if vp.get("fp64_flop_per_ray") else None,
                         "fp64_peak_tflops_unfused": 39.3,
                         "traffic": (vp["hbm_bytes_per_ray"] * rank_rays / launches) if vp.get("hbm_bytes_per_ray") is not None else vp.get("hbm_bytes_per_launch"),
                         "hbm_gbs": (vp["hbm_bytes_per_ray"] * rank_rays / k_s / 1e9) if vp.get("hbm_bytes_per_ray") is not None else None,
                         "hbm_frac_of_8TBs": (vp["hbm_bytes_per_ray"] * rank_rays / k_s / 8e12) if vp.get("hbm_bytes_per_ray") is not None else None,
                         "l2_hit_rate": vp.get("l2_hit_rate"), "pmc_spp": vp.get("pmc_spp"), "counters_measured_in_this_run": bool(vp.get("live")),
                         "wave_cycle_shares": vp.get("wave_cycle_shares"),
                         "per_ray_figures_from": vp.get("source"),
                         "note": ("per-ray instruction counts are rocprofv3 SQ counters of this kernel on this workload" +
                                  ("" if vp.get("pmc_spp") == args.spp else f" at {vp.get('pmc_spp')} spp (this run: {args.spp}: the 8-sample probe launch has another share of the frame)") +
                                  (", collected by child passes of this run before the timed loop; " if vp.get("live") else " (profiles/, not collected in this run); ")) +
                                 "rays and kernel time are this run's. frac = valu_issue_frac x lanes_active_frac at the nominal 2.4 GHz; the scene is LDS / L2-resident, "
                                 "HBM traffic is the framebuffer"})
        else:
            roof.update({"achieved": None, "frac": None, "note": "no profiles/valu.json for this workload: VALU figures not available"})
        # a rank whose shard has at most three pixels per lane renders its resume launch with the work-stealing instantiation:
        # taken-over pixels run through the cooperative body, whose instruction mix valu.json (profiled at N = 1) does not describe
        rank_pixels = rl.api.rows_for(H, row_first, row_step) * W
        if rank_pixels <= 3 * N_CU * 1024 and os.environ.get("RL_STEAL", "3") not in ("0", "0.0"):
            roof["kernel"] = "rtiow_wave_kernel<1024,4,false,true> (work stealing)"
            roof["note"] = (roof.get("note") or "") + "; this rank's shard is small enough for work stealing: the VALU figures are those of the N = 1 kernel and only indicative here"
        out = {
            "metric": "Mrays/sec (primary+secondary), 1080p 1024spp depth50; 1/2/4/8 GPU",
            "value": value, "unit": "Mrays/s", "n_gpus": G, "steps": args.steps, "warmup": args.warmup,
            "ms_per_step": ms_per_step, "higher_is_better": True, "scaling": "strong", "vs_baseline": None,
            "dtype": "f64", "data": "synthetic",
            "config": {"workload": f"RTIOW bouncing_spheres scene (488 spheres, 511-node BVH), {W}x{H}, {args.spp} spp, depth {args.depth}, seed 0",
                       "baseline_config": "configs[1]",
                       "sharding": (f"one process, rl_rtiow_render_multi_device over {G} GPU(s): rows interleaved, RCCL send/recv to GPU 0 inside the library" if args.inprocess
                                    else f"rows interleaved over {G} rank(s), RCCL gather to rank 0"),
                       "rays_per_step": rays, "aabb_tests_per_ray": nodes / rays, "sphere_tests_per_ray": spheres / rays},
            "roofline": roof,
        }
        if comm is not None:
            out["rccl"] = comm
        if args.emulate_shard > 1:
            out["config"]["emulated_shard_of"] = args.emulate_shard
        if rehearsal or (args.inprocess and os.environ.get("RL_BENCH_EMULATE_DEVICES")):
            out["config"]["rehearsal"] = "all ranks / device contexts on GPU 0: NOT a measurement"
        # ---- the check: the TIMED frame against the counting kernel's frame and against the CPU oracle at full spp
        if not args.no_check:
            timed_frame = frame if world_size > 1 else shard
            chk = {"timed_frame_equals_counting_frame": bool(torch.equal(timed_frame, counting_frame)),
                   "timed_rays_equal_counting_rays": timed_rays == rays, "flagged": int(st["flagged"])}
            full_rows = world_size > 1 or args.inprocess or args.emulate_shard <= 1  # timed_frame holds every image row
            rows = max(0, min(args.check_rows, H))
            if rows and full_rows:
                sys.path.insert(0, os.path.join(ROOT, "oracle"))
                oracle = importlib.import_module("rl_oracle")
                cstep = max(1, H // rows)
                cfirst = cstep // 2
                ys = np.arange(cfirst, H, cstep, dtype=np.uint32)
                # the same rows once more on the GPU as a row shard with the counting kernel: counters for exactly these rows
                rows_buf = torch.zeros((len(ys), W, 3), dtype=torch.float64, device=dev)
                rs = {}
                single = rl.World.bouncing_spheres(1) if args.inprocess else world
                if args.inprocess:
                    rl.init(0)  # back to one context for the row-shard render
                    single = rl.World.bouncing_spheres(1)
                cam.render_device(single, rows_buf.data_ptr(), stream=stream.cuda_stream, row_first=cfirst, row_step=cstep, stats=rs)
                torch.cuda.synchronize(dev)
                gx, gy = np.meshgrid(np.arange(W, dtype=np.uint32), ys)
                cs = {}
                c0 = time.perf_counter()
                cpu = oracle.rtiow_render_pixels(world.desc, cam.c, gx.ravel(), gy.ravel(), stats=cs).reshape(len(ys), W, 3)
                cdt = time.perf_counter() - c0
                timed_rows = timed_frame[torch.as_tensor(ys.astype(np.int64), device=dev)]
                chk.update({"rows_checked": int(len(ys)), "rows": [int(y) for y in ys], "pixels_checked": int(len(ys) * W), "spp": args.spp,
                            "timed_rows_equal_row_shard_render": bool(torch.equal(timed_rows, rows_buf)),
                            "counters_equal": all(int(rs[k]) == int(cs[k]) for k in ("rays", "node_tests", "sphere_tests", "rng_words", "flagged")),
                            "max_abs_err": float(np.abs(timed_rows.cpu().numpy() - cpu).max() / max(1, args.spp)),
                            "tolerance": 1e-4, "oracle_seconds": cdt})
            out["check"] = chk
        if world_size == 1 and not args.inprocess and not args.no_cpu_baseline:
            sys.path.insert(0, os.path.join(ROOT, "oracle"))
            oracle = importlib.import_module("rl_oracle")
            try:
                threads = min(oracle.hardware_threads(), len(os.sched_getaffinity(0)))
            except AttributeError:
                threads = oracle.hardware_threads()
            # bounded sample, sized for ~15 s of CPU work: every 2nd row, one task per PIXEL (no straggler rows), spp
            # calibrated from a 1-spp probe of the same pixels
            cstep = 2
            gx, gy = np.meshgrid(np.arange(W, dtype=np.uint32), np.arange(0, H, cstep, dtype=np.uint32))
            gx, gy = gx.ravel(), gy.ravel()
            probe = rl.Camera(rl.CameraParams(**{**p.__dict__, "samples_per_pixel": 1}))
            c0 = time.perf_counter()
            oracle.rtiow_render_pixels(world.desc, probe.c, gx, gy, threads=threads)
            pdt = max(time.perf_counter() - c0, 1e-3)
            cspp = int(max(1, min(args.spp, 15.0 / pdt)))
            ccam = rl.Camera(rl.CameraParams(**{**p.__dict__, "samples_per_pixel": cspp}))
            cst = {}
            c0 = time.perf_counter()
            oracle.rtiow_render_pixels(world.desc, ccam.c, gx, gy, threads=threads, stats=cst)
            cdt = time.perf_counter() - c0
            out["cpu_baseline"] = {"value": cst["rays"] / cdt / 1e6, "unit": "Mrays/s", "cores": threads, "kind": "port",
                                   "sample": f"same scene/camera {W}x{H} depth {args.depth}: rows y%{cstep}==0 ({rl.api.rows_for(H, 0, cstep)} rows), "
                                             f"{cspp} spp, {cst['rays']} rays in {cdt:.1f} s on {threads} threads (one task per pixel); CPU restatement of the "
                                             "reference algorithm (oracle/, -O3 -ffp-contract=off), not the Rust reference"}
        if world_size == 1 and not args.inprocess and args.emulate_shard <= 1 and args.configs:
            del shard, counting_frame
            out["configs"] = other_configs(args, rl, np, torch, dev)
        print(json.dumps(out), flush=True)
    if world_size > 1:
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
