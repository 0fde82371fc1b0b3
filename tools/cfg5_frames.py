import gzip, importlib, json, os, sys
import numpy as np, torch
ROOT=os.path.dirname(os.path.dirname(os.path.abspath(__file__))); sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "oracle"))
rl = importlib.import_module("rendering-learning_amd"); import rl_oracle as oracle
from PIL import Image
rl.init(0)
G = os.path.join(ROOT, "tests", "golden")
tex = np.asarray(Image.open(os.path.join(G, "spot_texture.png")).convert("RGB")); obj = gzip.open(os.path.join(G, "spot_triangulated.obj.gz"), "rb").read()
w = rl.World.stress_scene(1000, 2, obj, tex)
p = w.params; p.samples_per_pixel = int(sys.argv[1]) if len(sys.argv) > 1 else 4
if len(sys.argv) > 2: p.image_width = int(sys.argv[2])
cam = rl.Camera(p)
gs = {}
counting = cam.render(w, stats=gs).data
dev = torch.device("cuda", 0)
def timed():
    buf = torch.zeros((cam.c.image_height, cam.c.image_width, 3), dtype=torch.float64, device=dev)
    cam.render_device(w, buf.data_ptr(), stream=torch.cuda.current_stream(dev).cuda_stream)
    st = rl.api.render_status(w); return buf.cpu().numpy(), st
fast, st = timed()
rl.api.set_fast_traversal(False); ro, st0 = timed(); rl.api.set_fast_traversal(True)
print("rays counting", gs["rays"], "fast", st["rays"], "reforder", st0["rays"], "slow", st["slow_traces"])
for a, b, nm in ((fast, counting, "fast vs counting"), (ro, counting, "reforder vs counting"), (fast, ro, "fast vs reforder")):
    ys, xs = np.nonzero((a != b).any(axis=2)); print(nm, "differing pixels", len(ys), list(zip(xs[:8], ys[:8])))
ys, xs = np.nonzero((fast != counting).any(axis=2))
if len(ys):
    sel = slice(0, 64)
    cpu = oracle.rtiow_render_pixels(w.desc, cam.c, xs[sel].astype(np.uint32), ys[sel].astype(np.uint32))
    f = fast[ys[sel], xs[sel]]; c = counting[ys[sel], xs[sel]]
    print("oracle==fast", int((np.abs(cpu - f).max(axis=1) < 1e-9).sum()), "oracle==counting", int((np.abs(cpu - c).max(axis=1) < 1e-9).sum()), "of", len(cpu))
    print(cpu[:3], f[:3], c[:3])
