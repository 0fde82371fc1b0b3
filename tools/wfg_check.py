"""Wavefront form (rl_rtiow_wfg.h, RL_WAVEFRONT) against the megakernel and the counting kernel: bit equality of the frames, ray counts, timing.
usage: RL_RENDER_LIB=.../librl_render_exp.so wfg_check.py [small|cfg4|cfg5] [spp]   (the wavefront form is in the experimental library only)"""
import gzip, importlib, json, os, sys, time
import numpy as np, torch
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__))); sys.path.insert(0, ROOT)
rl = importlib.import_module("rendering-learning_amd")
from PIL import Image
G = os.path.join(ROOT, "tests", "golden")
which = sys.argv[1] if len(sys.argv) > 1 else "small"
spp = int(sys.argv[2]) if len(sys.argv) > 2 else 8
tex = np.asarray(Image.open(os.path.join(G, "spot_texture.png")).convert("RGB")); obj = gzip.open(os.path.join(G, "spot_triangulated.obj.gz"), "rb").read()
dev = torch.device("cuda", 0)
L = rl.api.render_lib()


def frame(w, p, variant, stats=None):
    rl.api.set_rtiow_variant(variant)
    cam = rl.Camera(p)
    buf = torch.zeros((cam.c.image_height, cam.c.image_width, 3), dtype=torch.float64, device=dev)
    s = torch.cuda.current_stream(dev)
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record(s)
    cam.render_device(w, buf.data_ptr(), stream=s.cuda_stream, stats=stats)
    e1.record(s)
    st = rl.api.render_status(w) if stats is None else stats
    torch.cuda.synchronize(dev)
    rl.api.set_rtiow_variant(0)
    return buf.cpu().numpy(), st, e0.elapsed_time(e1)


rl.init(0)
scenes = []
if which == "small":
    w = rl.World.cow_scene(obj, tex); p = w.params; p.image_width, p.samples_per_pixel = 200, spp
    scenes.append(("cow 200", w, p))
    w = rl.World.stress_scene(60, 1, obj, tex); p = w.params; p.image_width, p.samples_per_pixel = 256, spp
    scenes.append(("stress 60 / 256", w, p))
elif which == "cfg4":
    w = rl.World.cow_scene(obj, tex); p = w.params; p.aspect_ratio, p.image_width, p.samples_per_pixel = 16.0 / 9.0, 3840, spp
    scenes.append(("cfg4", w, p))
else:
    w = rl.World.stress_scene(1000, 2, obj, tex, device_bvh=True); p = w.params; p.samples_per_pixel = spp
    scenes.append(("cfg5", w, p))
for name, w, p in scenes:
    a, sa, ta = frame(w, p, 1031)
    a, sa, ta = frame(w, p, 1031)
    b, sb, tb = frame(w, p, 1035)
    b, sb, tb = frame(w, p, 1035)
    out = {"scene": name, "spp": p.samples_per_pixel, "mega_ms": ta, "wfg_ms": tb, "mega_Mrays_s": sa["rays"] / ta / 1e3, "wfg_Mrays_s": sb["rays"] / tb / 1e3,
           "rays_equal": sa["rays"] == sb["rays"], "frames_equal": bool(np.array_equal(a, b)), "differing_pixels": int((a != b).any(axis=2).sum()),
           "slow": [sa.get("slow_traces"), sb.get("slow_traces")]}
    if which == "small":
        st = {}
        c, _, tc = frame(w, p, 0, stats=st)
        out.update({"counting_equal": bool(np.array_equal(c, b)), "counting_rays_equal": st["rays"] == sb["rays"]})
    print(json.dumps(out), flush=True)
