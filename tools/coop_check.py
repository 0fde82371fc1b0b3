#!/usr/bin/env python3
"""A/B of the cooperative one-wave-per-pixel kernel (RL_RTIOW_KERNEL=coop / variant 1033: EVERY pixel through it) against the default
kernel on small images of the BASELINE scene: frames must be bit-equal; prints the time per ray of the longest chain."""
import ctypes as C, importlib, json, os, sys
import numpy as np, torch
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__))); sys.path.insert(0, ROOT)
rl = importlib.import_module("rendering-learning_amd"); rl.init(0)
dev = torch.device("cuda", 0)
L = rl.api.render_lib()
L.rl_debug_pixel_rays.argtypes = [C.c_void_p, C.c_uint64]; L.rl_debug_pixel_rays_read.argtypes = [C.c_void_p, C.c_void_p, C.c_uint64]
spp = int(sys.argv[1]) if len(sys.argv) > 1 else 256
sizes = [tuple(float(v) if i else int(v) for i, v in enumerate(a.split(':'))) for a in sys.argv[2:]] or [(1, 1.0), (8, 1.0), (32, 16.0 / 9.0), (128, 16.0 / 9.0)]
for width, aspect in sizes:
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
    out = {}
    for name, v in (("default", 0), ("coop", 1033)):
        rl.api.set_rtiow_variant(v)
        buf = torch.full((H, W, 3), float("nan"), dtype=torch.float64, device=dev)
        s = torch.cuda.current_stream(dev)
        best = None
        for _ in range(2):
            e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            e0.record(s); cam.render_device(world, buf.data_ptr(), stream=s.cuda_stream); e1.record(s)
            status = rl.api.render_status(world); torch.cuda.synchronize(dev)
            ms = e0.elapsed_time(e1); best = ms if best is None else min(best, ms)
        out[name] = (buf.cpu().numpy(), status, best)
    rl.api.set_rtiow_variant(0)
    same = bool(np.array_equal(out["default"][0], out["coop"][0]))
    print(json.dumps({"image": [W, H], "spp": spp, "frames_bit_equal": same, "rays": [int(out["default"][1]["rays"]), int(out["coop"][1]["rays"]), int(st["rays"])],
                      "slow_traces": [int(out["default"][1]["slow_traces"]), int(out["coop"][1]["slow_traces"])],
                      "ms": [out["default"][2], out["coop"][2]], "max_pixel_rays": int(rays.max()),
                      "us_per_ray_longest_chain": [out["default"][2] * 1e3 / max(1, int(rays.max())), out["coop"][2] * 1e3 / max(1, int(rays.max()))]}), flush=True)
    if not same:
        d = np.argwhere((out["default"][0] != out["coop"][0]).any(axis=2))
        print("first differing pixels", d[:5].tolist(), "of", len(d))
