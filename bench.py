#!/usr/bin/env python3
"""bench.py — BASELINE.json's headline metric on MI355X.

Metric: Mrays/sec (primary + secondary) on the RTIOW random-sphere scene
(examples/bouncing_spheres.rs: 488 spheres, 511-node BVH), 1920x1080, 1024 spp, depth 50 (configs[1]).
A "step" = one full render of that frame through the C ABI (rl_rtiow_render_device), scene already
resident in HBM, output left in HBM.  With N GPUs the framebuffer is sharded by interleaved rows
(row r -> rank r mod N, no data-path collective during the render) and gathered to rank 0 over
RCCL/xGMI once per step — inside the timed region.

Prints ONE JSON line (rank 0). Extra objects:
  roofline     — algorithmic bytes (64 B per AABB test + 64 B per sphere test + 208 B per ray, SURVEY.md
                 §8d, counted by the kernel itself and equal to the CPU oracle's counts) / kernel time,
                 against the 8 TB/s HBM peak.  The scene is LDS-resident, so the practical ceiling is
                 FP64 VALU issue under divergence, not HBM — see DESIGN.md.
  cpu_baseline — the CPU oracle (C++ restatement of the reference; the Rust reference cannot be built
                 here) timed on this box's host cores over a bounded sample of the same workload.
"""
import argparse
import importlib
import json
import os
import sys
import time

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)

HBM_PEAK_GBS = 8000.0  # /opt/skills/guides/MI355X_MICROARCH.md: HBM3E 8.0 TB/s spec


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=2)
    ap.add_argument("--warmup", type=int, default=1)
    ap.add_argument("--width", type=int, default=1920)
    ap.add_argument("--spp", type=int, default=1024)
    ap.add_argument("--depth", type=int, default=50)
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--emulate-shard", type=int, default=0,
                    help="single-GPU rehearsal of an N-way shard: render only rows 0 mod N (not the headline metric)")
    args = ap.parse_args()

    import torch
    import torch.distributed as dist

    world_size = int(os.environ.get("WORLD_SIZE", "1"))
    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    if world_size != args.gpus and world_size > 1:
        raise SystemExit(f"--gpus {args.gpus} but WORLD_SIZE={world_size}")
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

    rl = importlib.import_module("rendering-learning_amd")
    sharding = importlib.import_module("rendering-learning_amd.sharding")
    rl.init(local_rank)

    # ---- workload: configs[1] of BASELINE.json
    world = rl.World.bouncing_spheres(1)
    p = world.params
    p.image_width, p.samples_per_pixel, p.max_depth = args.width, args.spp, args.depth
    cam = rl.Camera(p)
    W, H = cam.c.image_width, cam.c.image_height
    G = world_size
    row_first, row_step = rank, G
    if args.emulate_shard > 1:
        row_first, row_step = 0, args.emulate_shard
    nrows = rl.api.rows_for(H, row_first, row_step)
    max_rows = rl.api.rows_for(H, 0, row_step)
    shard = torch.zeros((max_rows, W, 3), dtype=torch.float64, device=dev)
    gathered = [torch.zeros_like(shard) for _ in range(G)] if (G > 1 and rank == 0) else None
    frame = torch.zeros((H, W, 3), dtype=torch.float64, device=dev) if rank == 0 else None
    stream = torch.cuda.current_stream(dev)

    def exchange():  # the one exchange step of the path: framebuffer rows -> rank 0 over RCCL/xGMI
        if rehearsal:
            full = sharding.gather_frame(shard.cpu(), H, rank, G)
            if rank == 0:
                frame.copy_(full)
        else:
            sharding.gather_frame(shard, H, rank, G, frame=frame, gathered=gathered)

    def step(stats=None):
        cam.render_device(world, shard.data_ptr(), stream=stream.cuda_stream, row_first=row_first, row_step=row_step, stats=stats)
        if G > 1:
            exchange()

    # ---- counters for this exact workload (deterministic: identical for every launch) — untimed
    st = {}
    step(stats=st)
    torch.cuda.synchronize(dev)
    for _ in range(max(0, args.warmup - 1)):
        step()
    torch.cuda.synchronize(dev)
    if world_size > 1:
        dist.barrier()
    torch.cuda.synchronize(dev)

    evs = [(torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)) for _ in range(args.steps)]
    t0 = time.perf_counter()
    for k in range(args.steps):
        evs[k][0].record(stream)
        cam.render_device(world, shard.data_ptr(), stream=stream.cuda_stream, row_first=row_first, row_step=row_step)
        evs[k][1].record(stream)
        if G > 1:
            exchange()
    torch.cuda.synchronize(dev)
    if world_size > 1:
        dist.barrier()
    torch.cuda.synchronize(dev)
    elapsed = time.perf_counter() - t0
    kernel_ms = sum(a.elapsed_time(b) for a, b in evs) / max(1, args.steps)

    tot = torch.tensor([float(st["rays"]), float(st["node_tests"]), float(st["sphere_tests"]), elapsed, kernel_ms],
                       dtype=torch.float64, device="cpu" if rehearsal else dev)
    if world_size > 1:
        mx = tot.clone()
        dist.all_reduce(tot, op=dist.ReduceOp.SUM)
        dist.all_reduce(mx, op=dist.ReduceOp.MAX)
        elapsed = float(mx[3])
        kernel_ms_max = float(mx[4])
    else:
        kernel_ms_max = kernel_ms
    rays, nodes, spheres = float(tot[0]), float(tot[1]), float(tot[2])

    if rank == 0:
        ms_per_step = elapsed / args.steps * 1e3
        value = rays * args.steps / elapsed / 1e6
        # algorithmic bytes per launch on rank 0's kernel (per-rank share for N>1)
        alg_bytes = (64.0 * st["node_tests"] + 64.0 * st["sphere_tests"] + 208.0 * st["rays"])
        achieved = alg_bytes / (kernel_ms * 1e-3) / 1e9
        launches = 2 if (args.spp >= 64 and os.environ.get("RL_LPT", "1") != "0") else 1
        traffic, valu = None, None
        tpath = os.path.join(ROOT, "profiles", "traffic.json")
        if os.path.exists(tpath) and world_size == 1 and not args.emulate_shard:
            try:
                tj = json.load(open(tpath))
                if tj.get("workload") == f"{W}x{H}x{args.spp}spp_d{args.depth}":
                    traffic = tj.get("hbm_bytes_per_launch")
                    valu = tj.get("valu")  # SQ counters of the same kernel (profiles/): what actually binds it
            except Exception:
                traffic, valu = None, None
        out = {
            "metric": "Mrays/sec (primary+secondary), 1080p 1024spp depth50; 1/2/4/8 GPU",
            "value": value, "unit": "Mrays/s", "n_gpus": world_size, "steps": args.steps, "warmup": args.warmup,
            "ms_per_step": ms_per_step, "higher_is_better": True, "scaling": "strong", "vs_baseline": None,
            "dtype": "f64", "data": "synthetic",
            "config": {"workload": f"RTIOW bouncing_spheres scene (488 spheres, 511-node BVH), {W}x{H}, {args.spp} spp, depth {args.depth}, seed 0",
                       "baseline_config": "configs[1]", "sharding": f"rows interleaved over {world_size} rank(s), RCCL gather to rank 0",
                       "rays_per_step": rays, "aabb_tests_per_ray": nodes / rays, "sphere_tests_per_ray": spheres / rays},
            "roofline": {"bound": "hbm", "achieved": achieved, "peak": HBM_PEAK_GBS, "unit": "GB/s", "frac": achieved / HBM_PEAK_GBS,
                         "traffic": traffic, "kernel": "rtiow_wave_kernel", "kernel_ms": kernel_ms, "kernel_ms_max_rank": kernel_ms_max,
                         "algorithmic_bytes_per_launch": alg_bytes,
                         # spp >= 64: one render = two launches of the same kernel (8-sample cost probe + cost-sorted remainder);
                         # kernel_ms spans both, so rocprofv3's per-launch average = kernel_ms / launches_per_step
                         "launches_per_step": launches, "kernel_avg_launch_ms": kernel_ms / launches,
                         "valu_counters": valu,
                         "note": "scene is LDS / L2-resident: real HBM traffic is the framebuffer; the binding resource is VALU issue under divergence (valu_counters)"},
        }
        if args.emulate_shard > 1:
            out["config"]["emulated_shard_of"] = args.emulate_shard
        if rehearsal:
            out["config"]["rehearsal"] = "all ranks on GPU 0, gloo gather through host memory: NOT a measurement"
            # the assembled frame must equal a single-rank render of the same frame
            ref = torch.zeros((H, W, 3), dtype=torch.float64, device=dev)
            cam.render_device(world, ref.data_ptr(), stream=stream.cuda_stream)
            torch.cuda.synchronize(dev)
            out["config"]["rehearsal_frame_equal"] = bool(torch.equal(ref, frame))
        if world_size == 1 and not args.no_cpu_baseline:
            sys.path.insert(0, os.path.join(ROOT, "oracle"))
            oracle = importlib.import_module("rl_oracle")
            try:
                threads = min(oracle.hardware_threads(), len(os.sched_getaffinity(0)))
            except AttributeError:
                threads = oracle.hardware_threads()
            # bounded sample, sized for ~15 s of CPU work: every 2nd row (so every thread has rows), spp
            # calibrated from a 1-spp probe of the same rows
            cstep = 2
            probe = rl.Camera(rl.CameraParams(**{**p.__dict__, "samples_per_pixel": 1}))
            pst = {}
            c0 = time.perf_counter()
            oracle.rtiow_render(world.desc, probe.c, row_first=0, row_step=cstep, threads=threads, stats=pst)
            pdt = max(time.perf_counter() - c0, 1e-3)
            cspp = int(max(1, min(args.spp, 15.0 / pdt)))
            ccam = rl.Camera(rl.CameraParams(**{**p.__dict__, "samples_per_pixel": cspp}))
            cst = {}
            c0 = time.perf_counter()
            oracle.rtiow_render(world.desc, ccam.c, row_first=0, row_step=cstep, threads=threads, stats=cst)
            cdt = time.perf_counter() - c0
            out["cpu_baseline"] = {"value": cst["rays"] / cdt / 1e6, "unit": "Mrays/s", "cores": threads, "kind": "port",
                                   "sample": f"same scene/camera {W}x{H} depth {args.depth}: rows y%{cstep}==0 ({rl.api.rows_for(H, 0, cstep)} rows), "
                                             f"{cspp} spp, {cst['rays']} rays in {cdt:.1f} s on {threads} threads; CPU restatement of the "
                                             "reference algorithm (oracle/), not the Rust reference"}
        print(json.dumps(out), flush=True)
    if world_size > 1:
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
