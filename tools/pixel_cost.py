#!/usr/bin/env python3
"""Per-pixel ray counts of BASELINE configs[1] (debug output of the counting wave kernel): how long is the longest per-pixel
sample chain?  The strong-scaling floor of a 1/G shard is (rays of the most expensive pixel) x (latency per ray), because the
samples of one pixel are sequential (camera.rs:161-174).  usage: tools/pixel_cost.py [spp] [width]"""
import ctypes as C
import importlib
import json
import os
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
rl = importlib.import_module("rendering-learning_amd")
spp = int(sys.argv[1]) if len(sys.argv) > 1 else 64
width = int(sys.argv[2]) if len(sys.argv) > 2 else 1920
rl.init(0)
world = rl.World.bouncing_spheres(1)
p = world.params
p.image_width, p.samples_per_pixel, p.max_depth = width, spp, 50
cam = rl.Camera(p)
W, H = cam.c.image_width, cam.c.image_height
L = rl.api.render_lib()
L.rl_debug_pixel_rays.argtypes = [C.c_void_p, C.c_uint64]
L.rl_debug_pixel_rays_read.argtypes = [C.c_void_p, C.c_void_p, C.c_uint64]
assert L.rl_debug_pixel_rays(world.device(), W * H) == 0
st = {}
cam.render(world, stats=st)
rays = np.zeros(W * H, dtype=np.uint32)
assert L.rl_debug_pixel_rays_read(world.device(), rays.ctypes.data, W * H) == 0
rays = rays.reshape(H, W).astype(np.float64) / spp
assert abs(rays.sum() * spp - st["rays"]) < 0.5, (rays.sum() * spp, st["rays"])
flat = np.sort(rays.ravel())
q = {f"p{k}": float(flat[int(len(flat) * k / 100) - 1 if k else 0]) for k in (50, 90, 99)}
q["p99.9"], q["p99.99"], q["max"] = float(flat[int(len(flat) * 0.999)]), float(flat[int(len(flat) * 0.9999)]), float(flat[-1])
th, tw = H // 8, W // 8
tiles = rays[:th * 8, :tw * 8].reshape(th, 8, tw, 8).transpose(0, 2, 1, 3).reshape(th, tw, 64)
tmax, tmean = tiles.max(axis=2), tiles.mean(axis=2)
out = {"width": W, "height": H, "spp": spp, "rays_per_sample_mean": float(rays.mean()), "rays_per_sample_quantiles": q,
       "tile_mean_max": float(tmean.max()), "tile_max_over_mean_avg": float((tmax / np.maximum(tmean, 1e-9)).mean()),
       "argmax_pixel": [int(v) for v in np.unravel_index(np.argmax(rays), rays.shape)],
       "rows_max": [float(v) for v in np.sort(rays.max(axis=1))[-5:]],
       "kernel_ms": st["kernel_ms"], "rays": st["rays"]}
print(json.dumps(out))
np.save(os.path.join(ROOT, "gpurun_out", f"pixel_rays_{W}x{H}_{spp}.npy"), rays.astype(np.float32))
