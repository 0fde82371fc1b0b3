import importlib, os, sys, numpy as np, torch
sys.path.insert(0, "/root/repo")
rl = importlib.import_module("rendering-learning_amd"); rl.init(0)
dev = torch.device("cuda", 0)
world = rl.World.bouncing_spheres(1)
bad = 0
for (width, spp, shard) in ((640, 128, 1), (1920, 96, 8), (480, 256, 1), (1920, 64, 5)):
    p = world.params
    p.image_width, p.samples_per_pixel, p.max_depth = width, spp, 50
    cam = rl.Camera(p)
    rows = rl.api.rows_for(cam.c.image_height, 0, shard)
    buf = torch.zeros((rows, cam.c.image_width, 3), dtype=torch.float64, device=dev)
    s = torch.cuda.current_stream(dev)
    rl.api.set_steal(0.0)
    cam.render_device(world, buf.data_ptr(), stream=s.cuda_stream, row_first=0, row_step=shard); st0 = rl.api.render_status(world)
    ref = buf.cpu().numpy().copy()
    rl.api.set_steal(3.0)
    for it in range(12):
        buf.fill_(float("nan"))
        cam.render_device(world, buf.data_ptr(), stream=s.cuda_stream, row_first=0, row_step=shard); st = rl.api.render_status(world)
        ok = np.array_equal(buf.cpu().numpy(), ref) and st["rays"] == st0["rays"]
        bad += 0 if ok else 1
    print(width, spp, shard, "pixels", rows * cam.c.image_width, "mismatching runs so far", bad, flush=True)
print("TOTAL mismatching runs", bad)
