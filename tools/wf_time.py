import importlib, sys, time, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
rl = importlib.import_module("rendering-learning_amd")
rl.init(0)
w = rl.World.bouncing_spheres(1)
p = w.params; p.image_width=int(sys.argv[1]); p.samples_per_pixel=int(sys.argv[2]); p.max_depth=50
cam = rl.Camera(p)
for v in (0,3):
    rl.api.set_rtiow_variant(v)
    st={}; cam.render(w, stats=st)
    t=time.time(); st={}; cam.render(w, stats=st); dt=time.time()-t
    print("variant", v, "kernel_ms", round(st["kernel_ms"],1), "wall_ms", round(dt*1e3,1), "Mrays/s", round(st["rays"]/st["kernel_ms"]/1e3,1), flush=True)
