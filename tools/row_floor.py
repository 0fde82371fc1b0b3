import importlib, sys, os, time
sys.path.insert(0, "/root/repo")
import numpy as np
rl = importlib.import_module("rendering-learning_amd"); rl.init(0)
w = rl.World.bouncing_spheres(1); p = w.params
p.image_width, p.samples_per_pixel, p.max_depth = 1920, 1024, 50
cam = rl.Camera(p)
for r in (300, 500, 600, 650, 700, 750, 800, 900):
    st = {}
    cam.render_rows(w, r, 100000, stats=st)
    print("row", r, "rays", st["rays"], "rays/sample/pixel %.2f" % (st["rays"] / 1920 / 1024), "kernel_ms %.1f" % st["kernel_ms"], flush=True)
