"""GPU tier: renders of ONE scene issued concurrently — from several host threads, on several HIP streams — as the reference's
Camera::render(&self, world) allows (ray-tracing-one-weekend/src/camera.rs:122: `&self`, `world: H + Sync`).  The scene owns one set of work
buffers; the library serialises the host side and orders the renders on the device (csrc/rl_scene.h rl_scene::mu), so every frame must be
the bits of the same render issued alone, and rl_render_status must account for every render (rays of the latest one, every flag)."""
import threading

import numpy as np
import pytest

pytestmark = pytest.mark.gpu


def test_concurrent_renders_of_one_scene_from_threads_and_streams(rl):
    import torch
    dev = torch.device("cuda", 0)
    world = rl.World.bouncing_spheres(1)
    cams = []
    for k, (w, spp, seed) in enumerate([(96, 70, 0), (64, 6, 1), (128, 3, 2), (80, 66, 3)]):  # both sides of the two-launch threshold
        p = rl.CameraParams(**{**world.params.__dict__, "image_width": w, "samples_per_pixel": spp, "max_depth": 20, "seed": seed})
        cams.append(rl.Camera(p))
    alone = []
    for cam in cams:
        gs = {}
        alone.append((cam.render(world, stats=gs).data, gs["rays"]))
    streams = [torch.cuda.Stream(dev) for _ in cams]
    for rounds in range(3):
        bufs = [torch.full((c.c.image_height, c.c.image_width, 3), float("nan"), dtype=torch.float64, device=dev) for c in cams]
        errors = []

        def work(i):
            try:
                for _ in range(3):  # each thread re-renders its frame: 12 renders in flight against a status ring of 8
                    cams[i].render_device(world, bufs[i].data_ptr(), stream=streams[i].cuda_stream)
            except Exception as e:  # noqa: BLE001
                errors.append((i, repr(e)))
        ts = [threading.Thread(target=work, args=(i,)) for i in range(len(cams))]
        for t in ts:
            t.start()
        for t in ts:
            t.join()
        assert not errors, errors
        st = rl.api.render_status(world)
        torch.cuda.synchronize(dev)
        for i, cam in enumerate(cams):
            assert np.array_equal(bufs[i].cpu().numpy(), alone[i][0]), (rounds, i)
        assert st["rays"] in [a[1] for a in alone] and st["flagged"] == 0  # the rays of whichever render was enqueued last


def test_concurrent_rtc_renders_of_one_world(rl, golden):
    import torch
    dev = torch.device("cuda", 0)
    w = rl.RtcWorld.test_obj_scene(golden("teapot-low.obj"), 120, 80)
    ref1, ref2 = w.render(1), w.render(2)
    s1, s2 = torch.cuda.Stream(dev), torch.cuda.Stream(dev)
    a = torch.full((80, 120, 3), float("nan"), dtype=torch.float64, device=dev)
    b = torch.full_like(a, float("nan"))
    t1 = threading.Thread(target=lambda: [w.render_device(a.data_ptr(), 1, stream=s1.cuda_stream) for _ in range(4)])
    t2 = threading.Thread(target=lambda: [w.render_device(b.data_ptr(), 2, stream=s2.cuda_stream) for _ in range(4)])
    t1.start(), t2.start(), t1.join(), t2.join()
    rl.api.render_status(w)
    torch.cuda.synchronize(dev)
    assert np.array_equal(a.cpu().numpy(), ref1) and np.array_equal(b.cpu().numpy(), ref2)


def test_progress_of_a_running_render_can_be_polled(rl):
    """rl_rtiow_render_progress — the C ABI's form of camera.rs:176-184 ("Scanline-equivalents remaining", once per image_width finished
    pixels): polled from the host while an asynchronous render runs, it must not wait for the render and must climb to the launch's total."""
    import time

    import torch
    dev = torch.device("cuda", 0)
    world = rl.World.bouncing_spheres(1)
    assert rl.api.render_progress(world) == (0, 0, 0)  # the first call switches the host-visible work counters on
    p = rl.CameraParams(**{**world.params.__dict__, "image_width": 1280, "samples_per_pixel": 256, "max_depth": 50})
    cam = rl.Camera(p)
    buf = torch.zeros((cam.c.image_height, cam.c.image_width, 3), dtype=torch.float64, device=dev)
    cam.render_device(world, buf.data_ptr(), stream=torch.cuda.current_stream(dev).cuda_stream)
    seen, t0 = [], time.perf_counter()
    while time.perf_counter() - t0 < 20.0:
        claimed, total, phase = rl.api.render_progress(world)
        seen.append((claimed, total, phase))
        if phase == 1 and claimed >= total:
            break
        time.sleep(0.001)
    st = rl.api.render_status(world)
    gs = {}
    assert np.array_equal(buf.cpu().numpy(), cam.render(world, stats=gs).data) and st["rays"] == gs["rays"]  # same frame with the counters in host memory
    tiles = ((cam.c.image_width + 7) // 8) * ((cam.c.image_height + 7) // 8)
    assert all(t == tiles * 64 for _, t, _ in seen)
    assert seen[-1][0] == tiles * 64 and seen[-1][2] == 1 and st["rays"] > 0
    mid = [c for c, t, _ in seen if 0 < c < t]
    assert mid, seen[:5]  # the poll returned while the kernels were still handing pixels out
    for ph in (0, 1):  # within a launch the count only grows
        cs = [c for c, _, q in seen if q == ph]
        assert cs == sorted(cs)
