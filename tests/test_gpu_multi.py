"""GPU tier: the one-process multi-GPU entry points of the C ABI (rl_init_multi / rl_*_render_multi) on the
one-GPU test box: n_devices = 1, and G emulated device contexts on the same GPU (own stream, scene replica and shard buffer per
context, rows g, g+G, ... per context, device-to-device exchange, de-interleave kernel) asserting bit-equality with the single
render.  What a one-GPU box cannot run is the RCCL send / recv itself (a communicator needs distinct devices); the exchange here
is the peer-copy path.  This file runs in a subprocess so that the re-initialised library does not leak into the other tests."""
import os
import subprocess
import sys

import pytest

pytestmark = pytest.mark.gpu
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))

BODY = r'''
import importlib, os, sys
import numpy as np
import torch  # BEFORE the product library: torch bundles its own HIP runtime, and a process must not end up with two of them
sys.path.insert(0, %(root)r)
rl = importlib.import_module("rendering-learning_amd")
api = rl.api

def golden(name):
    import gzip
    p = os.path.join(%(root)r, "tests", "golden", name)
    return gzip.open(p, "rb").read() if name.endswith(".gz") else open(p, "rb").read()

# ---- single context first: the reference frames
api.init(0)
assert api.render_lib().rl_device_count() == 1
world = rl.World.bouncing_spheres(1)
p = world.params
p.image_width, p.samples_per_pixel, p.max_depth = 101, 66, 50   # 101 x 56: ragged against every G; 66 spp: two-launch path
cam = rl.Camera(p)
gs = {}
single = cam.render(world, stats=gs).data
rw = rl.RtcWorld.test_obj_scene(golden("teapot-low.obj"), 90, 61)
rs = {}
rsingle = rw.render(2, stats=rs)
mirror = rl.RtcWorld.test_mirror_scene(75, 50)
msingle = mirror.render(1)
# n_devices = 1 through the multi entry point
ms = {}
assert np.array_equal(cam.render_multi(world, stats=ms).data, single)
assert all(ms[k] == gs[k] for k in ("rays", "node_tests", "sphere_tests", "rng_words", "flagged"))
del world, rw, mirror

for G in (2, 3, 8):
    assert api.init_multi(emulate=G) == G
    world = rl.World.bouncing_spheres(1)       # re-created: one replica per context
    ms = {}
    frame = cam.render_multi(world, stats=ms).data
    assert np.array_equal(frame, single), G
    assert all(ms[k] == gs[k] for k in ("rays", "node_tests", "sphere_tests", "rng_words", "flagged")), (G, ms, gs)
    # asynchronous form: frame left on device 0, completion through rl_render_status
    buf = torch.full((cam.c.image_height, cam.c.image_width, 3), float("nan"), dtype=torch.float64, device="cuda:0")
    torch.cuda.synchronize()
    cam.render_multi_device(world, buf.data_ptr())
    st = api.render_status(world)
    assert np.array_equal(buf.cpu().numpy(), single) and st["rays"] == gs["rays"], G
    # two asynchronous frames back to back, no rl_render_status in between: the second frame's shards must not land in the gather slots
    # before the first frame's de-interleave has read them (ADVICE r02: the peer-copy gather used to race here)
    quick = rl.Camera(rl.CameraParams(**{**p.__dict__, "samples_per_pixel": 3, "seed": 9}))
    quick_ref = None
    for attempt in range(3):
        a = torch.full((cam.c.image_height, cam.c.image_width, 3), float("nan"), dtype=torch.float64, device="cuda:0")
        b = torch.full_like(a, float("nan"))
        torch.cuda.synchronize()
        cam.render_multi_device(world, a.data_ptr())
        quick.render_multi_device(world, b.data_ptr())
        api.render_status(world)
        if quick_ref is None:
            quick_ref = quick.render_multi(world).data
        assert np.array_equal(a.cpu().numpy(), single), (G, attempt)
        assert np.array_equal(b.cpu().numpy(), quick_ref), (G, attempt)
    rw = rl.RtcWorld.test_obj_scene(golden("teapot-low.obj"), 90, 61)
    rms = {}
    assert np.array_equal(rw.render_multi(2, stats=rms), rsingle), G
    assert all(rms[k] == rs[k] for k in ("rays", "node_tests", "planar_tests", "instance_enters", "flagged")), G
    mirror = rl.RtcWorld.test_mirror_scene(75, 50)
    assert np.array_equal(mirror.render_multi(1), msingle), G
    del world, rw, mirror
# more contexts than image rows: the surplus ranks render nothing
assert api.init_multi(emulate=8) == 8
tiny = rl.CameraParams(**{**p.__dict__, "image_width": 9, "samples_per_pixel": 4})   # 9 x 5
api.init(0)
w1 = rl.World.bouncing_spheres(1)
ref = rl.Camera(tiny).render(w1).data
del w1
api.init_multi(emulate=8)
w8 = rl.World.bouncing_spheres(1)
assert np.array_equal(rl.Camera(tiny).render_multi(w8).data, ref)
# real rl_init_multi on however many GPUs this box has (1 here): loads nothing it does not need and renders the same frame
del w8
n = api.init_multi(0)
assert n >= 1
wn = rl.World.bouncing_spheres(1)
assert np.array_equal(cam.render_multi(wn).data, single)
print("MULTI_OK", n, int(api.render_lib().rl_debug_multi_uses_rccl()))
'''


def test_multi_gpu_entry_points_emulated_row_groups_bit_equal_single_render():
    r = subprocess.run([sys.executable, "-c", BODY % {"root": ROOT}], stdout=subprocess.PIPE, stderr=subprocess.STDOUT, text=True, timeout=900)
    assert r.returncode == 0 and "MULTI_OK" in r.stdout, r.stdout[-4000:]
