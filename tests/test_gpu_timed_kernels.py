"""GPU tier: the kernel instantiations bench.py TIMES — `rl_*_render_device(..., stats = NULL)`, i.e. the STATS=false code
objects (different register allocation and spill pattern from the counting ones every other test runs) — compared bit for bit
with the counting instantiation's frame and against the oracle.  Covers rtiow_wave_kernel<1024,3,false> (compact guarded ops),
<1024,2,false> (64-byte linked ops), the LDS / HBM fallbacks, rtiow_wave_general_kernel<512,{false,true},false>, the
cost-sorted two-launch (LPT) path (spp >= 64), rtc_kernel / rtc_full_kernel, and the Flat material (material.rs:52-67).
"""
import gzip
import os

import numpy as np
import pytest

pytestmark = pytest.mark.gpu
TOL = 1e-4
COUNTERS = ("rays", "node_tests", "sphere_tests", "planar_tests", "instance_enters", "rng_words", "flagged")


def _device_frame(rl, cam, world, first_sample=0, row_first=0, row_step=1):
    """rl_rtiow_render_device with stats = NULL into a torch buffer on the current stream (what bench.py does)."""
    import torch
    dev = torch.device("cuda", 0)
    nrows = rl.api.rows_for(cam.c.image_height, row_first, row_step)
    buf = torch.full((nrows, cam.c.image_width, 3), float("nan"), dtype=torch.float64, device=dev)
    stream = torch.cuda.current_stream(dev)
    cam.render_device(world, buf.data_ptr(), stream=stream.cuda_stream, row_first=row_first, row_step=row_step, first_sample=first_sample)
    st = rl.api.render_status(world)  # synchronises; rays / flagged of the asynchronous render
    return buf.cpu().numpy(), st


def _timed_vs_counting_vs_oracle(rl, oracle, world, p, tight=1e-9):
    cam = rl.Camera(p)
    gs, cs = {}, {}
    counting = cam.render(world, stats=gs).data
    timed, st = _device_frame(rl, cam, world)
    assert np.array_equal(timed, counting)  # bit-identical: same arithmetic, same order, other code object
    assert st["rays"] == gs["rays"] and st["flagged"] == gs["flagged"]
    cpu = oracle.rtiow_render(world.desc, cam.c, stats=cs)
    for k in COUNTERS:
        assert gs[k] == cs[k], (k, gs[k], cs[k])
    assert np.abs(timed - cpu).max() / max(1, p.samples_per_pixel) <= TOL
    assert np.abs(timed - cpu).max() <= tight * max(1.0, np.abs(cpu).max())
    return gs


@pytest.mark.parametrize("variant", [0, 1025, 1027, 1024, 1029, 1033])
@pytest.mark.parametrize("spp", [6, 72])
def test_timed_sphere_kernels_equal_counting_kernels_and_oracle(rl, oracle, variant, spp):
    """BASELINE configs[1] scene; spp = 72 takes the cost-sorted two-launch path at its real threshold (>= 64)."""
    world = rl.World.bouncing_spheres(1)
    p = world.params
    p.image_width, p.samples_per_pixel, p.max_depth = 128, spp, 50
    try:
        # 0 = the default (counting: rtiow_wave_kernel<1024, 3, true>; timed: the cooperative kernel for a frame this small);
        # 1029 = the fast traversal <1024, 4, false> (what larger frames use), 1033 = the cooperative one-wave-per-pixel kernel
        rl.api.set_rtiow_variant(variant)
        _timed_vs_counting_vs_oracle(rl, oracle, world, p)
    finally:
        rl.api.set_rtiow_variant(0)


def test_timed_kernel_row_shards_and_resume(rl, oracle):
    world = rl.World.bouncing_spheres(1)
    p = world.params
    p.image_width, p.samples_per_pixel, p.max_depth = 96, 66, 50
    cam = rl.Camera(p)
    full = cam.render(world).data
    for g, G in ((0, 2), (1, 2), (5, 8)):
        part, _ = _device_frame(rl, cam, world, row_first=g, row_step=G)
        assert np.array_equal(part, full[g::G]), (g, G)
    resumed, _ = _device_frame(rl, cam, world, first_sample=66)
    cpu = oracle.rtiow_render(world.desc, cam.c, first_sample=66)
    assert np.abs(resumed - cpu).max() <= 1e-9 * max(1.0, np.abs(cpu).max())


def _spot_texture():
    from PIL import Image
    root = os.path.dirname(os.path.abspath(__file__))
    return np.asarray(Image.open(os.path.join(root, "golden", "spot_texture.png")).convert("RGB"))


@pytest.mark.parametrize("spp", [6, 64])
def test_timed_general_kernel_cow_scene(rl, oracle, golden, spp):
    """rtiow_wave_general_kernel<512, false, false>: BASELINE configs[3] scene (no transcendental textures), reduced."""
    world = rl.World.cow_scene(golden("spot_triangulated.obj.gz"), _spot_texture())
    p = world.params
    p.aspect_ratio, p.image_width, p.samples_per_pixel = 16.0 / 9.0, 96, spp
    gs = _timed_vs_counting_vs_oracle(rl, oracle, world, p)
    assert gs["planar_tests"] > 0 and gs["instance_enters"] > 0


def test_timed_general_kernel_with_noise_and_sphere_uv(rl, oracle):
    """rtiow_wave_general_kernel<512, true, false>: Noise (sin / Perlin) and Image-on-sphere (acos / atan2) textures."""
    tex = _spot_texture()
    lin = (tex.astype(np.float32) / 255.0) ** 2.2

    def scene(b):
        n1 = b.lambertian(b.noise(4.0, 7))
        img = b.lambertian(b.image(lin))
        objs = [b.sphere((0, -1000, 0), 1000, n1), b.sphere((0, 2, 0), 2, img), b.sphere((3, 1, 2), 1, b.dielectric(1.5)),
                b.quad((-4, 0, -3), (3, 0, 0), (0, 3, 0), b.diffuse_light(b.solid((3, 3, 3))))]
        return b.bvh(objs)
    world = rl.World.build(scene)
    p = rl.CameraParams(aspect_ratio=1.5, image_width=96, samples_per_pixel=6, max_depth=12, vfov=30.0, lookfrom=(13, 3, 5), lookat=(0, 1, 0))
    _timed_vs_counting_vs_oracle(rl, oracle, world, p)


def _flat_scene(b):
    """material.rs:52-67 `Flat`: never scatters, emits nothing -> a hit ends the path with (0, 0, 0)."""
    flat = b.flat()
    return [b.sphere((0, 0, -1), 0.5, flat), b.sphere((1.1, 0, -1.2), 0.5, b.metal((0.9, 0.9, 0.9), 0.0)),
            b.sphere((-1.1, 0, -1.2), 0.5, b.dielectric(1.5)), b.sphere((0, -100.5, -1), 100, b.lambertian(b.solid((0.6, 0.6, 0.2)))),
            b.sphere((0.2, 1.3, -1.5), 0.4, flat, center2=(0.2, 1.6, -1.5))]


@pytest.mark.parametrize("general", [False, True])
def test_flat_material_on_the_gpu(rl, oracle, general):
    def scene(b):
        objs = _flat_scene(b)
        if general:  # a Flat quad too: the all-primitives kernel
            objs.append(b.quad((-2, -0.5, -3), (4, 0, 0), (0, 2.5, 0), b.flat()))
        return b.bvh(objs)
    world = rl.World.build(scene)
    p = rl.CameraParams(aspect_ratio=1.5, image_width=96, samples_per_pixel=8, max_depth=10, vfov=60.0, lookfrom=(0, 0.3, 1.5), lookat=(0, 0, -1),
                        background=(0.7, 0.8, 1.0), seed=11)
    gs = _timed_vs_counting_vs_oracle(rl, oracle, world, p)
    cam = rl.Camera(rl.CameraParams(**{**p.__dict__, "samples_per_pixel": 1, "max_depth": 1}))
    one = cam.render(world).data  # depth 1: a primary ray that hits the Flat sphere in the image centre returns black
    assert not one[32, 48].any() and one[0, 0].any()
    assert gs["rays"] > 96 * 64 * 8


def test_timed_rtc_kernels_equal_counting_frames(rl, oracle, golden):
    import torch
    dev = torch.device("cuda", 0)
    for world in (rl.RtcWorld.test_obj_scene(golden("teapot-low.obj"), 150, 100), rl.RtcWorld.test_mirror_scene(150, 100), rl.RtcWorld.test_csg_scene(150, 100)):
        gs = {}
        counting = world.render(2, stats=gs)
        buf = torch.full((100, 150, 3), float("nan"), dtype=torch.float64, device=dev)
        world.render_device(buf.data_ptr(), aa_samples=2, stream=torch.cuda.current_stream(dev).cuda_stream)
        st = rl.api.render_status(world)
        assert np.array_equal(buf.cpu().numpy(), counting)
        assert st["rays"] == gs["rays"] and st["flagged"] == gs["flagged"]
        cpu = oracle.rtc_render(world.desc, world.camera, aa=2)
        assert np.abs(counting - cpu).max() <= 1e-9


def test_async_render_surfaces_reference_panic_sites(rl, oracle):
    """rl_render_status after an asynchronous render reports RL_E_DEGENERATE although no stats were requested: a unit sphere
    at x = 1e12 makes `(p - center) / radius` miss unit length by ~1e-4 (p carries 1e-16 * 1e12 of rounding), which trips
    the from_normalized assert of vec3.rs:219 for thousands of hits."""
    import torch
    api = rl.api
    tex = np.zeros(1, dtype=api.TEXTURE)
    tex[0]["kind"], tex[0]["color"] = api.TEX_SOLID, (0.5, 0.4, 0.3)
    mats = np.zeros(1, dtype=api.MATERIAL)
    mats[0]["kind"], mats[0]["texture"] = api.MAT_LAMBERTIAN, 0
    sph = np.zeros(1, dtype=api.SPHERE)
    sph[0]["center0"], sph[0]["radius"] = (1e12, 0.0, -1.0), 1.0
    world = rl.World.from_spheres(sph, mats, tex, False)
    p = rl.CameraParams(aspect_ratio=1.0, image_width=33, samples_per_pixel=16, max_depth=5, lookfrom=(1e12, 0, 3), lookat=(1e12, 0, -1), vfov=40.0)
    cam = rl.Camera(p)
    gs, cs = {}, {}
    cam.render(world, stats=gs, allow_degenerate=True)
    oracle.rtiow_render(world.desc, cam.c, stats=cs)
    assert gs["flagged"] == cs["flagged"] > 1000 and gs["rc"] == api.RL_E_DEGENERATE
    buf = torch.zeros((33, 33, 3), dtype=torch.float64, device="cuda:0")
    cam.render_device(world, buf.data_ptr(), stream=torch.cuda.current_stream().cuda_stream)
    st = api.render_status(world, allow_degenerate=True)
    assert st["rc"] == api.RL_E_DEGENERATE and st["flagged"] == gs["flagged"] and st["rays"] == gs["rays"]
    with pytest.raises(api.RLError):
        cam.render_device(world, buf.data_ptr(), stream=torch.cuda.current_stream().cuda_stream)
        api.render_status(world)
