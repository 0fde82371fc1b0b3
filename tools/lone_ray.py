#!/usr/bin/env python3
"""Latency of a lone sample chain: renders tiny images of the BASELINE scene (1x1, 8x8 = one wave, 64x36 = 36 waves on 256 CUs) with the
timed kernels (the wave-scheduled fast kernel, and the cooperative one-wave-per-pixel kernel that small frames use by default) at 1024 spp
on an otherwise idle GPU and divides the kernel time by the rays of the most expensive pixel (counting render).
This is the per-ray latency that bounds a 1/G shard: shard time >= rays of the worst pixel x this latency."""
import ctypes as C
import importlib
import json
import os
import sys

import numpy as np
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
rl = importlib.import_module("rendering-learning_amd")
rl.init(0)
L = rl.api.render_lib()
L.rl_debug_pixel_rays.argtypes = [C.c_void_p, C.c_uint64]
L.rl_debug_pixel_rays_read.argtypes = [C.c_void_p, C.c_void_p, C.c_uint64]
dev = torch.device("cuda", 0)
spp = int(sys.argv[1]) if len(sys.argv) > 1 else 1024
for width, aspect in ((1, 1.0), (2, 1.0), (8, 1.0), (64, 16.0 / 9.0), (256, 16.0 / 9.0)):
    world = rl.World.bouncing_spheres(1)
    p = world.params
    p.image_width, p.aspect_ratio, p.samples_per_pixel, p.max_depth = width, aspect, spp, 50
    cam = rl.Camera(p)
    W, H = cam.c.image_width, cam.c.image_height
    assert L.rl_debug_pixel_rays(world.device(), W * H) == 0
    st = {}
    cam.render(world, stats=st)
    rays = np.zeros(W * H, dtype=np.uint32)
    assert L.rl_debug_pixel_rays_read(world.device(), rays.ctypes.data, W * H) == 0
    buf = torch.zeros((H, W, 3), dtype=torch.float64, device=dev)
    stream = torch.cuda.current_stream(dev)
    res = {}
    for name, coop in (("wave_scheduled_kernel", False), ("cooperative_kernel", True)):  # the latter is the default for frames this small
        rl.api.set_coop(coop)
        best = None
        for _ in range(3):
            e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            e0.record(stream)
            cam.render_device(world, buf.data_ptr(), stream=stream.cuda_stream)
            e1.record(stream)
            rl.api.render_status(world)
            torch.cuda.synchronize(dev)
            ms = e0.elapsed_time(e1)
            best = ms if best is None else min(best, ms)
        res[name] = {"timed_ms": best, "us_per_ray_of_the_longest_chain": best * 1e3 / max(1, int(rays.max()))}
    rl.api.set_coop(True)
    print(json.dumps({"image": [W, H], "spp": spp, "pixels": W * H, "rays": int(rays.sum()), "max_pixel_rays": int(rays.max()), **res,
                      "counting_kernel_ms": st["kernel_ms"]}), flush=True)
