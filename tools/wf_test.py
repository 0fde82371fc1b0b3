import importlib, sys, numpy as np, time
sys.path.insert(0,'/root/repo'); sys.path.insert(0,'/root/repo/oracle')
rl = importlib.import_module("rendering-learning_amd"); import rl_oracle as o
rl.init(0)
w = rl.World.bouncing_spheres(1)
p = w.params; p.image_width=200; p.samples_per_pixel=8; p.max_depth=50
cam = rl.Camera(p)
gs={}; a = cam.render(w, stats=gs).data
rl.api.set_rtiow_variant(3)
g3={}; b = cam.render(w, stats=g3).data
print("equal", np.array_equal(a,b), "maxdiff", np.abs(a-b).max())
for k in ("rays","node_tests","sphere_tests","rng_words","flagged"): print(k, gs[k], g3[k])
wg = rl.World.golden_test_scene(); cg = rl.Camera(wg.params)
c3 = cg.render(wg).data; rl.api.set_rtiow_variant(0); c0 = cg.render(wg).data
print("golden equal", np.array_equal(c3,c0))
# timing at 1080p 32 spp
p.image_width=1920; p.samples_per_pixel=32
cam = rl.Camera(p)
for v in (0,3):
    rl.api.set_rtiow_variant(v)
    st={}; cam.render(w, stats=st); st={}; cam.render(w, stats=st)
    print("variant", v, "kernel_ms", round(st["kernel_ms"],1), "Mrays/s", round(st["rays"]/st["kernel_ms"]/1e3,1))
