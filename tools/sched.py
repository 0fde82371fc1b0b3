"""Prints the wave scheduler's block executions / average population per state (STATS launch)."""
import ctypes as C, importlib, sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
rl = importlib.import_module("rendering-learning_amd")
rl.init(0)
w = rl.World.bouncing_spheres(1)
p = w.params; p.image_width = int(sys.argv[1]) if len(sys.argv) > 1 else 1920; p.samples_per_pixel = int(sys.argv[2]) if len(sys.argv) > 2 else 32; p.max_depth = 50
cam = rl.Camera(p)
if os.environ.get("RL_SCHED_FAST", "1") != "0":  # the timed kernel's fast traversal, instrumented (its box / sphere counts are its own)
    rl.api.render_lib().rl_debug_fast_stats(1)
st = {}
cam.render(w, stats=st)
out = (C.c_uint64 * 32)()
L = rl.api.render_lib(); L.rl_debug_sched.argtypes = [C.c_void_p, C.c_void_p]
L.rl_debug_sched(w.device(), out)
names = {0: "GEN", 1: "TRAV", 2: "SHADE", 3: "FILL", 5: "LEAF", 6: "SHADE2"}
print("rays", st["rays"], "kernel_ms", st["kernel_ms"], "Mrays/s", st["rays"] / st["kernel_ms"] / 1e3, "box tests/ray", st["node_tests"] / st["rays"], "sphere tests/ray", st["sphere_tests"] / st["rays"])
tot = sum(out[3 * k + 2] for k in names)
for k, nm in names.items():
    ex, pop, cyc = out[3 * k], out[3 * k + 1], out[3 * k + 2]
    if ex or cyc: ex = max(ex, 1); print(f"{nm:6s} execs {ex:12d}  lanes served {pop:14d}  avg pop {pop/ex:6.2f}  lane-visits/ray {pop/st['rays']:.3f}  "
                 f"cycles/exec {cyc/ex:8.1f}  time share {100*cyc/tot:5.1f}%")
