"""ConstantMedium (hittable/constant_medium.rs:9-80) + Isotropic (material.rs:197-219), deterministic variant: the free path is
drawn from the pixel's ChaCha8 stream instead of the process-global rand::random (constant_medium.rs:55), see include/rl_render.h
rl_medium.  CPU tier: the oracle against hand-computed draws and the Beer-Lambert law; GPU tier: device = oracle."""
import math

import numpy as np
import pytest


def _slab_world(rl, density, boundary="sphere"):
    def scene(b):
        fog = b.isotropic(b.solid((1.0, 1.0, 1.0)))
        if boundary == "sphere":
            shape = b.sphere((0, 0, -5), 1.0, b.flat())
        else:  # an axis-aligned box of six quads, 2 x 2 x 2, centred at (0, 0, -5)
            m = b.flat()
            shape = b.list([b.quad((-1, -1, -4), (2, 0, 0), (0, 2, 0), m), b.quad((-1, -1, -6), (2, 0, 0), (0, 2, 0), m),
                            b.quad((-1, -1, -6), (0, 0, 2), (0, 2, 0), m), b.quad((1, -1, -6), (0, 0, 2), (0, 2, 0), m),
                            b.quad((-1, -1, -6), (2, 0, 0), (0, 0, 2), m), b.quad((-1, 1, -6), (2, 0, 0), (0, 0, 2), m)])
        return b.list([b.constant_medium(shape, density, fog)])
    return rl.World.build(scene)


def test_oracle_medium_first_draw_by_hand(rl, oracle):
    """depth 1: a scattered ray contributes 0, a miss the background.  Per sample the draws are px, py, time (camera.rs:203-216), then the
    medium's free path -(1/density) * ln(gen::<f64>()) — the fourth f64 of the pixel's stream for sample 0."""
    density = 0.4
    world = _slab_world(rl, density)
    p = rl.CameraParams(aspect_ratio=1.0, image_width=9, samples_per_pixel=1, max_depth=1, vfov=10.0, lookfrom=(0, 0, 0), lookat=(0, 0, -5),
                        background=(0.25, 0.5, 0.75), seed=3)
    cam = rl.Camera(p)
    img = oracle.rtiow_render(world.desc, cam.c)
    W = H = 9
    x = y = 4  # the centre pixel looks through ~2 units of fog (a little less off-axis: recompute the chord from the actual ray)
    ops = [("set_stream", 0 * W * H + x * W + y), ("f64",), ("f64",), ("f64",), ("f64",)]
    out, _ = oracle.chacha_script(p.seed, ops)
    px, py, _time, u = out[1], out[2], out[3], out[4]
    c = cam.c
    centre = np.array(c.pixel_00) + np.array(c.pixel_du) * x + np.array(c.pixel_dv) * y
    sample = centre + np.array(c.pixel_du) * (-0.5 + px) + np.array(c.pixel_dv) * (-0.5 + py)
    d = sample - np.array(c.lookfrom)
    oc = np.array(c.lookfrom) - np.array([0.0, 0.0, -5.0])
    a, hb, cc = d @ d, oc @ d, oc @ oc - 1.0
    sq = math.sqrt(hb * hb - a * cc)
    inside = ((-hb + sq) / a - (-hb - sq) / a) * math.sqrt(a)
    hit_distance = -(1.0 / density) * math.log(u)
    want = np.zeros(3) if hit_distance <= inside else np.array(p.background)
    assert np.array_equal(img[y, x], want)
    corner = img[0, 0]  # misses the sphere: no draw, background
    assert np.array_equal(corner, np.array(p.background))


@pytest.mark.parametrize("boundary", ["sphere", "box"])
def test_oracle_medium_obeys_beer_lambert(rl, oracle, boundary):
    density = 0.35
    world = _slab_world(rl, density, boundary)
    p = rl.CameraParams(aspect_ratio=1.0, image_width=3, samples_per_pixel=20000, max_depth=1, vfov=0.5, lookfrom=(0, 0, 0), lookat=(0, 0, -5),
                        background=(1.0, 1.0, 1.0), seed=11)
    cam = rl.Camera(p)
    img = oracle.rtiow_render(world.desc, cam.c) / p.samples_per_pixel
    want = math.exp(-density * 2.0)  # a 2-unit chord along the axis (vfov 0.5 degrees: every ray is practically axial)
    sigma = math.sqrt(want * (1 - want) / p.samples_per_pixel)
    assert abs(img[1, 1, 0] - want) < 5 * sigma
    again = oracle.rtiow_render(world.desc, cam.c) / p.samples_per_pixel  # deterministic: same seed, same image
    assert np.array_equal(img, again)
    other = oracle.rtiow_render(world.desc, rl.Camera(rl.CameraParams(**{**p.__dict__, "seed": 12})).c) / p.samples_per_pixel
    assert not np.array_equal(img, other)


def _smoke_scene(b):
    """examples/cornell_smoke.rs in miniature: a room of quads, a light, two rotated / translated boxes of smoke, plus final_scene.rs's
    glass ball with a medium inside."""
    white = b.lambertian(b.solid((0.73, 0.73, 0.73)))
    green = b.lambertian(b.solid((0.12, 0.45, 0.15)))
    red = b.lambertian(b.solid((0.65, 0.05, 0.05)))
    light = b.diffuse_light(b.solid((7, 7, 7)))

    def box(lo, hi, m):
        (x0, y0, z0), (x1, y1, z1) = lo, hi
        dx, dy, dz = (x1 - x0, 0, 0), (0, y1 - y0, 0), (0, 0, z1 - z0)
        return b.list([b.quad((x0, y0, z1), dx, dy, m), b.quad((x1, y0, z1), (0, 0, -(z1 - z0)), dy, m), b.quad((x1, y0, z0), (-(x1 - x0), 0, 0), dy, m),
                       b.quad((x0, y0, z0), dz, dy, m), b.quad((x0, y1, z1), dx, (0, 0, -(z1 - z0)), m), b.quad((x0, y0, z0), dx, dz, m)])
    room = [b.quad((555, 0, 0), (0, 555, 0), (0, 0, 555), green), b.quad((0, 0, 0), (0, 555, 0), (0, 0, 555), red),
            b.quad((113, 554, 127), (330, 0, 0), (0, 0, 305), light), b.quad((0, 555, 0), (555, 0, 0), (0, 0, 555), white),
            b.quad((0, 0, 0), (555, 0, 0), (0, 0, 555), white), b.quad((0, 0, 555), (555, 0, 0), (0, 555, 0), white)]
    box1 = b.translate(b.rotate_y(box((0, 0, 0), (165, 330, 165), white), 15.0), (265, 0, 295))
    box2 = b.translate(b.rotate_y(box((0, 0, 0), (165, 165, 165), white), -18.0), (130, 0, 65))
    smoke1 = b.constant_medium(box1, 0.01, b.isotropic(b.solid((0, 0, 0))))
    smoke2 = b.constant_medium(box2, 0.01, b.isotropic(b.solid((1, 1, 1))))
    ball = b.sphere((400, 90, 120), 60, b.dielectric(1.5))
    haze = b.constant_medium(b.sphere((400, 90, 120), 60, b.dielectric(1.5)), 0.2, b.isotropic(b.solid((0.2, 0.4, 0.9))))
    return b.bvh(room + [smoke1, smoke2, ball, haze])


@pytest.mark.gpu
def test_gpu_constant_medium_equals_oracle(rl, oracle):
    import torch
    world = rl.World.build(_smoke_scene)
    p = rl.CameraParams(aspect_ratio=1.0, image_width=72, samples_per_pixel=24, max_depth=30, vfov=40.0, lookfrom=(278, 278, -800), lookat=(278, 278, 0),
                        background=(0, 0, 0), seed=5)
    cam = rl.Camera(p)
    gs, cs = {}, {}
    gpu = cam.render(world, stats=gs).data
    cpu = oracle.rtiow_render(world.desc, cam.c, stats=cs)
    for k in ("rays", "node_tests", "sphere_tests", "planar_tests", "instance_enters", "rng_words", "flagged"):
        assert gs[k] == cs[k], (k, gs[k], cs[k])
    # colours: log() is the device libm here and glibc in the oracle (<= 2 ulp apart); it feeds the scatter POINT, i.e. colour-level noise
    assert np.abs(gpu - cpu).max() <= 1e-9 * max(1.0, np.abs(cpu).max())
    buf = torch.full((cam.c.image_height, cam.c.image_width, 3), float("nan"), dtype=torch.float64, device="cuda:0")
    cam.render_device(world, buf.data_ptr(), stream=torch.cuda.current_stream().cuda_stream)
    st = rl.api.render_status(world)
    assert np.array_equal(buf.cpu().numpy(), gpu) and st["rays"] == gs["rays"] and st["slow_traces"] == 0
    assert cam.render(world).data.tobytes() == gpu.tobytes()  # deterministic


@pytest.mark.gpu
def test_gpu_medium_row_shards_and_first_sample(rl, oracle):
    world = _slab_world(rl, 0.4, "box")
    p = rl.CameraParams(aspect_ratio=1.5, image_width=48, samples_per_pixel=9, max_depth=8, vfov=35.0, lookfrom=(0.5, 0.4, 0), lookat=(0, 0, -5),
                        background=(0.7, 0.8, 1.0), seed=2)
    cam = rl.Camera(p)
    full = cam.render(world).data
    for g, G in ((0, 2), (1, 2), (2, 5)):
        assert np.array_equal(cam.render_rows(world, g, G), full[g::G])
    part = cam.render_rows(world, 0, 1, first_sample=9)
    cpu = oracle.rtiow_render(world.desc, cam.c, first_sample=9)
    assert np.abs(part - cpu).max() <= 1e-9


@pytest.mark.gpu
@pytest.mark.parametrize("spp", [6, 72])
def test_gpu_wave_scheduled_medium_scope_equals_nested_loop_kernel(rl, oracle, spp):
    """The wave-scheduled kernel evaluates a ConstantMedium as a scope of the threaded program (park the closest hit, walk the boundary
    twice, draw the free path); the nested-loop kernel (RL_RTIOW_KERNEL=general) does it by recursion.  Same frames, same counters —
    also through the cost-sorted two-launch render (spp 72) and for the counter-free instantiation."""
    import torch
    world = rl.World.build(_smoke_scene)
    p = rl.CameraParams(aspect_ratio=1.0, image_width=64, samples_per_pixel=spp, max_depth=20, vfov=40.0, lookfrom=(278, 278, -800), lookat=(278, 278, 0),
                        background=(0.02, 0.02, 0.03), seed=11)
    cam = rl.Camera(p)
    frames, stats = {}, {}
    try:
        for v in (0, 2):  # 0: default (wave-scheduled, variant 4 for scenes with media), 2: nested loops
            rl.api.set_rtiow_variant(v)
            st = {}
            frames[v] = cam.render(world, stats=st).data
            stats[v] = st
            buf = torch.full((cam.c.image_height, cam.c.image_width, 3), float("nan"), dtype=torch.float64, device="cuda:0")
            cam.render_device(world, buf.data_ptr(), stream=torch.cuda.current_stream().cuda_stream)
            ts = rl.api.render_status(world)
            assert np.array_equal(buf.cpu().numpy(), frames[v]) and ts["rays"] == st["rays"], v
    finally:
        rl.api.set_rtiow_variant(0)
    assert np.array_equal(frames[0], frames[2])
    for k in ("rays", "node_tests", "sphere_tests", "planar_tests", "instance_enters", "rng_words", "flagged"):
        assert stats[0][k] == stats[2][k], (k, stats[0][k], stats[2][k])
    if spp == 6:
        cs = {}
        cpu = oracle.rtiow_render(world.desc, cam.c, stats=cs)
        assert cs["rays"] == stats[0]["rays"] and cs["rng_words"] == stats[0]["rng_words"]
        assert np.abs(frames[0] - cpu).max() <= 1e-9 * max(1.0, np.abs(cpu).max())


def _boundary_shapes_scene(b):
    """Media behind every kind of boundary the fast traversal tells apart (rl_program.h FastMedium::shape): a sphere, a moving sphere under a
    Translate, a rotated box of six quads, a tetrahedron of triangles, THREE coincident boxes in one list (six hits per line, the three
    nearest equal: the one-pass evaluation must hand over to the reference's fold), a Bvh of spheres (general shape), and a medium whose
    boundary is an open surface (one quad: never two hits, never a draw)."""
    white = b.lambertian(b.solid((0.73, 0.73, 0.73)))
    fog = lambda c: b.isotropic(b.solid(c))
    flat = b.flat()

    def box(lo, hi, m):
        (x0, y0, z0), (x1, y1, z1) = lo, hi
        dx, dy, dz = (x1 - x0, 0, 0), (0, y1 - y0, 0), (0, 0, z1 - z0)
        return [b.quad((x0, y0, z1), dx, dy, m), b.quad((x1, y0, z1), (0, 0, -(z1 - z0)), dy, m), b.quad((x1, y0, z0), (-(x1 - x0), 0, 0), dy, m),
                b.quad((x0, y0, z0), dz, dy, m), b.quad((x0, y1, z1), dx, (0, 0, -(z1 - z0)), m), b.quad((x0, y0, z0), dx, dz, m)]
    floor = b.quad((-20, -1, -20), (40, 0, 0), (0, 0, 40), white)
    lamp = b.quad((-3, 8, -3), (6, 0, 0), (0, 0, 6), b.diffuse_light(b.solid((6, 6, 6))))
    m_sphere = b.constant_medium(b.sphere((-6, 0.5, 0), 1.5, flat), 0.5, fog((0.9, 0.2, 0.2)))
    m_moving = b.constant_medium(b.translate(b.sphere((0, 0, 0), 1.2, flat, center2=(0.4, 0.3, 0)), (-2.5, 0.5, 1.0)), 0.7, fog((0.2, 0.9, 0.2)))
    m_box = b.constant_medium(b.translate(b.rotate_y(b.list(box((0, 0, 0), (2, 2.5, 2), flat)), 25.0), (0.5, -1, -1)), 0.6, fog((0.2, 0.2, 0.9)))
    tet = [(4, -1, -1), (6, -1, -1), (5, -1, 1), (5, 1.5, 0)]
    tri = lambda i, j, k: b.triangle(tet[i], tuple(np.subtract(tet[j], tet[i])), tuple(np.subtract(tet[k], tet[i])), flat)
    m_tet = b.constant_medium(b.list([tri(0, 1, 2), tri(0, 1, 3), tri(1, 2, 3), tri(2, 0, 3)]), 0.9, fog((0.9, 0.9, 0.2)))
    m_triple = b.constant_medium(b.list(box((7, -1, -1), (9, 1, 1), flat) * 3), 0.8, fog((0.2, 0.9, 0.9)))
    m_general = b.constant_medium(b.bvh([b.sphere((-4, 3.5, -2), 1.0, flat), b.sphere((-3, 3.5, -2), 1.0, flat), b.sphere((-2, 3.5, -2), 1.0, flat)]), 0.8,
                                  fog((0.9, 0.2, 0.9)))
    m_open = b.constant_medium(b.quad((2, 2, -2), (2, 0, 0), (0, 2, 0), flat), 5.0, fog((1, 1, 1)))
    ball = b.sphere((2.5, 3.5, 1.0), 0.8, b.dielectric(1.5))
    return b.list([floor, m_sphere, lamp, m_moving, b.bvh([m_box, ball, m_tet]), m_triple, m_general, m_open])


@pytest.mark.gpu
def test_gpu_fast_traversal_with_every_boundary_shape_equals_counting_kernel_and_oracle(rl, oracle):
    import torch
    world = rl.World.build(_boundary_shapes_scene)
    p = rl.CameraParams(aspect_ratio=1.6, image_width=96, samples_per_pixel=20, max_depth=12, vfov=45.0, lookfrom=(1, 4, 16), lookat=(1, 1, 0),
                        background=(0.3, 0.4, 0.6), seed=17)
    cam = rl.Camera(p)
    out16 = (rl.api.C.c_uint64 * 16)()
    rl.api.render_lib().rl_debug_host_structures.argtypes = [rl.api.C.c_void_p, rl.api.C.c_void_p]
    assert rl.api.render_lib().rl_debug_host_structures(world.desc, out16) == 0 and (out16[0] & 2)  # the general fast structure exists
    gs, cs = {}, {}
    counting = cam.render(world, stats=gs).data
    buf = torch.full((cam.c.image_height, cam.c.image_width, 3), float("nan"), dtype=torch.float64, device="cuda:0")
    cam.render_device(world, buf.data_ptr(), stream=torch.cuda.current_stream().cuda_stream)
    st = rl.api.render_status(world)
    fast = buf.cpu().numpy()
    assert np.array_equal(fast.view(np.uint64), counting.view(np.uint64)), int((fast != counting).any(axis=2).sum())
    assert st["rays"] == gs["rays"] and st["flagged"] == gs["flagged"]
    cpu = oracle.rtiow_render(world.desc, cam.c, stats=cs)
    for k in ("rays", "node_tests", "sphere_tests", "planar_tests", "instance_enters", "rng_words", "flagged"):
        assert gs[k] == cs[k], (k, gs[k], cs[k])
    assert np.abs(counting - cpu).max() <= 1e-9 * max(1.0, np.abs(cpu).max())


def _planes_scene(b):
    """Unbounded Planes (plane.rs: no interior test, AABB::universe) everywhere the fast traversal has to cope with them: in a list, inside
    a Bvh (where they drag every enclosing box to the universe), under a Translate / rotation, between two media (so that their segment
    matters), and as part of a medium's boundary (fog between two planes: no box node, every ray evaluates it)."""
    grey = b.lambertian(b.checker(0.8, b.solid((0.2, 0.3, 0.1)), b.solid((0.9, 0.9, 0.9))))
    red = b.lambertian(b.solid((0.8, 0.2, 0.2)))
    flat = b.flat()
    ground = b.plane((0, -1, 0), (1, 0, 0), (0, 0, 1), grey)
    wall = b.translate(b.rotate_y(b.plane((0, 0, -6), (1, 0, 0), (0, 1, 0), red), 20.0), (0.5, 0, 0))
    mirror = b.plane((-7, 0, 0), (0, 0, 1), (0, 1, 0), b.metal((0.8, 0.8, 0.9), 0.05))
    lamp = b.quad((-2, 5, -2), (4, 0, 0), (0, 0, 4), b.diffuse_light(b.solid((5, 5, 5))))
    balls = [b.sphere((x, -0.4, z), 0.6, m) for x, z, m in ((-2, 0, b.dielectric(1.5)), (0, -1, red), (2, 0.5, b.metal((0.7, 0.6, 0.5), 0.0)))]
    slab = b.constant_medium(b.list([b.plane((0, 0.2, 0), (1, 0, 0), (0, 0, 1), flat), b.plane((0, 0.9, 0), (1, 0, 0), (0, 0, 1), flat)]), 0.15,
                             b.isotropic(b.solid((0.8, 0.8, 1.0))))
    puff = b.constant_medium(b.sphere((3, 1.5, -2), 1.0, flat), 0.8, b.isotropic(b.solid((1.0, 0.6, 0.2))))
    return b.list([ground, b.bvh(balls + [wall, lamp]), puff, mirror, slab, b.sphere((-3, 2, -3), 0.7, red)])


@pytest.mark.gpu
def test_gpu_fast_traversal_with_unbounded_planes_equals_counting_kernel_and_oracle(rl, oracle):
    import torch
    world = rl.World.build(_planes_scene)
    out16 = (rl.api.C.c_uint64 * 16)()
    rl.api.render_lib().rl_debug_host_structures.argtypes = [rl.api.C.c_void_p, rl.api.C.c_void_p]
    assert rl.api.render_lib().rl_debug_host_structures(world.desc, out16) == 0 and (out16[0] & 2)
    assert out16[14] == 2 and out16[15] == 0x0012  # puff: a sphere behind a box node; slab: general shape, no box node
    for seed, depth in ((3, 1), (4, 12)):
        p = rl.CameraParams(aspect_ratio=1.5, image_width=90, samples_per_pixel=16, max_depth=depth, vfov=60.0, lookfrom=(0.5, 1.5, 9), lookat=(0, 0.3, 0),
                            background=(0.4, 0.5, 0.8), seed=seed)
        cam = rl.Camera(p)
        gs, cs = {}, {}
        counting = cam.render(world, stats=gs).data
        buf = torch.full((cam.c.image_height, cam.c.image_width, 3), float("nan"), dtype=torch.float64, device="cuda:0")
        cam.render_device(world, buf.data_ptr(), stream=torch.cuda.current_stream().cuda_stream)
        st = rl.api.render_status(world)
        fast = buf.cpu().numpy()
        assert np.array_equal(fast.view(np.uint64), counting.view(np.uint64)), (depth, int((fast != counting).any(axis=2).sum()))
        assert st["rays"] == gs["rays"] and st["flagged"] == gs["flagged"]
        cpu = oracle.rtiow_render(world.desc, cam.c, stats=cs)
        for k in ("rays", "node_tests", "sphere_tests", "planar_tests", "instance_enters", "rng_words", "flagged"):
            assert gs[k] == cs[k], (k, gs[k], cs[k])
        assert np.abs(counting - cpu).max() <= 1e-9 * max(1.0, np.abs(cpu).max())
