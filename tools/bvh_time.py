import importlib, sys, time, os
sys.path.insert(0, "/root/repo")
import numpy as np
rl = importlib.import_module("rendering-learning_amd"); rl.init(0)
for dev in (False, True, True):
    t0 = time.perf_counter()
    w = rl.World.stress_scene(1000, 0, None, None, device_bvh=dev)
    print("device_bvh", dev, "1M-sphere scene build %.2f s" % (time.perf_counter() - t0), flush=True)
p = w.params; p.image_width, p.samples_per_pixel = 480, 2
st = {}
rl.Camera(p).render(w, stats=st)
print({k: st[k] for k in ("rays", "node_tests", "sphere_tests")})
