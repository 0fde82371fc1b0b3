"""GPU tier: the edge cases of the hot path — empty and ragged inputs, degenerate sizes, the reference's panic sites —
through the C ABI against the oracle (same bars as test_gpu_parity.py: counters exact, colours <= 1e-4)."""
import numpy as np
import pytest

pytestmark = pytest.mark.gpu
TOL = 1e-4
COUNTERS = ("rays", "node_tests", "sphere_tests", "planar_tests", "instance_enters", "rng_words", "flagged")


def _check(rl, oracle, world, p, allow_degenerate=False):
    cam = rl.Camera(p)
    gs, cs = {}, {}
    gpu = cam.render(world, stats=gs, allow_degenerate=allow_degenerate).data
    cpu = oracle.rtiow_render(world.desc, cam.c, stats=cs)
    for k in COUNTERS:
        assert gs[k] == cs[k], (k, gs[k], cs[k])
    assert gpu.shape == cpu.shape
    if gpu.size:
        fin = np.isfinite(cpu)
        assert np.array_equal(np.isfinite(gpu), fin)
        assert np.abs(gpu[fin] - cpu[fin]).max(initial=0.0) <= TOL * max(1, p.samples_per_pixel)
    return gs, gpu


def _materials(rl):
    api = rl.api
    tex = np.zeros(1, dtype=api.TEXTURE)
    tex[0]["kind"], tex[0]["color"] = api.TEX_SOLID, (0.5, 0.4, 0.3)
    mats = np.zeros(3, dtype=api.MATERIAL)
    mats[0]["kind"], mats[0]["texture"] = api.MAT_LAMBERTIAN, 0
    mats[1]["kind"], mats[1]["ior"] = api.MAT_DIELECTRIC, 1.5
    mats[2]["kind"], mats[2]["albedo"], mats[2]["fuzz"] = api.MAT_METAL, (0.9, 0.9, 0.9), 0.0
    return tex, mats


@pytest.mark.parametrize("use_bvh", [False])
def test_empty_world_is_all_background(rl, oracle, use_bvh):
    tex, mats = _materials(rl)
    world = rl.World.from_spheres(np.zeros(0, dtype=rl.api.SPHERE), mats, tex, use_bvh)
    p = rl.CameraParams(aspect_ratio=2.0, image_width=37, samples_per_pixel=3, max_depth=5, background=(0.25, 0.5, 0.75))
    gs, gpu = _check(rl, oracle, world, p)
    assert gs["rays"] == 37 * 18 * 3 and gs["node_tests"] == 0 and gs["sphere_tests"] == 0
    assert np.array_equal(gpu, np.broadcast_to(np.array([0.25, 0.5, 0.75]) * 3, gpu.shape))


@pytest.mark.parametrize("w,ar,spp,depth", [(1, 1.0, 1, 1), (1, 1.0, 5, 50), (7, 7.0, 2, 3), (9, 0.25, 1, 2), (65, 1.0, 1, 50), (8, 1.0, 64, 2), (13, 1.3, 70, 4)])
def test_ragged_image_sizes_sample_counts_and_depths(rl, oracle, w, ar, spp, depth):
    """widths / heights that are not multiples of the 8x8 claim tile, a single pixel, one sample, spp on both sides of the
    64-sample threshold of the cost-sorted two-launch render, depth 1 (every path ends after its first scatter)."""
    world = rl.World.golden_test_scene()
    p = world.params
    p.aspect_ratio, p.image_width, p.samples_per_pixel, p.max_depth = ar, w, spp, depth
    _check(rl, oracle, world, p)


def test_zero_depth_and_zero_samples(rl, oracle):
    world = rl.World.golden_test_scene()
    p = world.params
    p.image_width, p.samples_per_pixel, p.max_depth = 24, 4, 0  # ray_color(depth 0) = black (camera.rs:233): no ray is traced
    gs, gpu = _check(rl, oracle, world, p)
    assert gs["rays"] == 0 and not gpu.any()
    p.samples_per_pixel, p.max_depth = 0, 10  # no samples: the canvas is all zeros
    gs, gpu = _check(rl, oracle, world, p)
    assert gs["rays"] == 0 and not gpu.any()


def test_first_sample_offsets_and_row_shards_of_a_ragged_image(rl, oracle):
    world = rl.World.bouncing_spheres(1)
    p = world.params
    p.image_width, p.samples_per_pixel, p.max_depth = 61, 3, 8
    cam = rl.Camera(p)
    for first, (g, G) in ((0, (0, 1)), (1000, (2, 5)), (2**33, (33, 34)), (7, (40, 41))):
        gpu = cam.render_rows(world, g, G, first_sample=first)
        cpu = oracle.rtiow_render(world.desc, cam.c, first_sample=first, row_first=g, row_step=G)
        assert gpu.shape == cpu.shape
        if gpu.size:
            assert np.abs(gpu - cpu).max() <= 1e-9
    assert cam.render_rows(world, 500, 3).shape[0] == 0  # first row past the image: nothing to do, no error


def test_degenerate_inputs_reach_the_reference_panic_sites_as_flags(rl, oracle):
    """A sphere of radius 0 makes `(p - center) / radius` non-finite: NormalizedVec3::from_normalized would panic
    (vec3.rs:219).  The library counts it (rl_stats.flagged, RL_E_DEGENERATE) instead of unwinding; the counters still
    match the oracle's."""
    api = rl.api
    tex, mats = _materials(rl)
    sph = np.zeros(2, dtype=api.SPHERE)
    sph["center0"] = [(0, 0, -1), (0.6, 0, -1)]
    sph["radius"] = [0.5, 0.0]
    sph["material"] = [0, 0]
    world = rl.World.from_spheres(sph, mats, tex, False)
    p = rl.CameraParams(aspect_ratio=1.0, image_width=32, samples_per_pixel=2, max_depth=4, lookfrom=(0, 0, 1), lookat=(0, 0, -1))
    cam = rl.Camera(p)
    gs, cs = {}, {}
    cam.render(world, stats=gs, allow_degenerate=True)
    oracle.rtiow_render(world.desc, cam.c, stats=cs)
    for k in COUNTERS:
        assert gs[k] == cs[k], (k, gs[k], cs[k])


def test_rays_parallel_to_box_faces_and_huge_scenes_take_the_exact_path(rl, oracle):
    """Axis-parallel rays (a zero direction component: the filtered AABB test must fall back to the reference's divisions,
    +-inf and NaN included) and coordinates of 1e150 (the filter's magnitude guard)."""
    api = rl.api
    tex, mats = _materials(rl)
    sph = np.zeros(6, dtype=api.SPHERE)
    sph["center0"] = [(0, 0, -2), (1, 0, -2), (-1, 0, -2), (0, 1, -2), (0, -1, -2), (0, 0, -1e150)]
    sph["radius"] = [0.5, 0.5, 0.5, 0.5, 0.5, 9e149]
    sph["material"] = [0, 1, 2, 0, 2, 0]
    world = rl.World.from_spheres(sph, mats, tex, True)
    # camera looking exactly down -z from the axis: the centre column / row of rays has d.x == 0 or d.y == 0
    p = rl.CameraParams(aspect_ratio=1.0, image_width=33, samples_per_pixel=2, max_depth=6, vfov=60.0, lookfrom=(0, 0, 2), lookat=(0, 0, -2),
                        defocus_angle=0.0)
    _check(rl, oracle, world, p, allow_degenerate=True)


def test_directions_and_origins_outside_the_binary32_range_take_the_exact_path(rl, oracle):
    """The headline kernel filters AABB tests in binary32: a direction component of 1e-50 (1/d overflows a float), one of
    1e45, or an origin 1e35 away must switch the lane to the reference's binary64 divisions, not to inf / NaN arithmetic."""
    api = rl.api
    tex, mats = _materials(rl)
    rng = np.random.default_rng(3)
    sph = np.zeros(40, dtype=api.SPHERE)
    sph["center0"] = rng.uniform(-3, 3, (40, 3)) + (0, 0, -6)
    sph["radius"] = rng.uniform(0.2, 0.9, 40)
    sph["material"] = rng.integers(0, 3, 40)
    world = rl.World.from_spheres(sph, mats, tex, True)
    # vfov tiny + huge focus distance: pixel_du / pixel_dv of ~1e-50 (tiny x / y direction components after lookfrom - pixel)
    for kw in (dict(vfov=1e-48, lookfrom=(0.0, 0.0, 2.0), lookat=(0.0, 0.0, -6.0)),
               dict(vfov=40.0, lookfrom=(0.0, 0.0, 1e35), lookat=(0.0, 0.0, -6.0)),
               dict(vfov=40.0, lookfrom=(1e-45, 2e-45, 2.0), lookat=(0.0, 0.0, -6.0))):
        p = rl.CameraParams(aspect_ratio=1.0, image_width=24, samples_per_pixel=2, max_depth=5, defocus_angle=0.0, **kw)
        _check(rl, oracle, world, p, allow_degenerate=True)


def test_rtc_empty_world_no_lights_and_single_pixel(rl, oracle):
    api = rl.api
    empty = rl.RtcWorld.from_arrays(np.zeros(0, dtype=api.RTC_TRIANGLE), np.zeros(0, dtype=api.RTC_MATERIAL), np.zeros(0, dtype=api.HREF),
                                    np.zeros(0, dtype=api.RTC_LIGHT), void_color=(0.1, 0.2, 0.3))
    cam = rl.rtc_camera(5, 3, 1.0, (0, 0, -5), (0, 0, 0), (0, 1, 0))
    img = empty.render(aa_samples=2, camera=cam)
    assert img.shape == (3, 5, 3) and np.allclose(img, (0.1, 0.2, 0.3), atol=1e-15)
    assert np.array_equal(img, oracle.rtc_render(empty.desc, cam, aa=2))
    # one triangle, no lights: shade_hit sums over zero lights = black where hit, void elsewhere (world.rs:71-90)
    t = np.zeros(1, dtype=api.RTC_TRIANGLE)
    p1, p2, p3 = np.array([0, 1, 0.0]), np.array([-1, 0, 0.0]), np.array([1, 0, 0.0])
    t["p1"], t["e1"], t["e2"], t["n1"] = p1, p2 - p1, p3 - p1, (0, 0, -1)
    m = np.zeros(1, dtype=api.RTC_MATERIAL)
    m["color"], m["ambient"], m["diffuse"], m["specular"], m["shininess"], m["refractive_index"] = (1, 1, 1), 0.1, 0.9, 0.9, 200.0, 1.0
    objs = np.zeros(1, dtype=api.HREF)
    objs["kind"], objs["index"] = api.O_TRIANGLE, 0
    dark = rl.RtcWorld.from_arrays(t, m, objs, np.zeros(0, dtype=api.RTC_LIGHT), void_color=(0.1, 0.2, 0.3))
    for (w, h) in ((1, 1), (9, 7)):
        cam = rl.rtc_camera(w, h, 1.0, (0, 0.4, -5), (0, 0.4, 0), (0, 1, 0))
        gs, cs = {}, {}
        a = dark.render(aa_samples=3, camera=cam, stats=gs)
        b = oracle.rtc_render(dark.desc, cam, aa=3, stats=cs)
        assert np.abs(a - b).max() <= 1e-12
        for k in ("rays", "node_tests", "sphere_tests"):
            assert gs[k] == cs[k]


def test_rtc_triangle_guard_tree_keeps_order_ties_and_axis_parallel_rays(rl, oracle):
    """The fast RTC kernel walks a reject-only box tree over triangle ranges of >= 8 triangles.  Coincident triangles (ties in t:
    the later one must win, its material shows), triangles lying in axis planes (flat boxes), rays parallel to an axis
    (1/d = inf: the tree must never reject) and shadow rays all have to come out as in the reference's plain loop."""
    api = rl.api
    rng = np.random.default_rng(17)
    n = 40
    tri = np.zeros(n, dtype=api.RTC_TRIANGLE)
    p1 = rng.uniform(-2, 2, (n, 3))
    e1, e2 = rng.uniform(-1.5, 1.5, (n, 3)), rng.uniform(-1.5, 1.5, (n, 3))
    p1[:8, 2], e1[:8, 2], e2[:8, 2] = 1.0, 0.0, 0.0           # eight triangles in the plane z = 1 (zero-thickness boxes)
    p1[20:24], e1[20:24], e2[20:24] = p1[8:12], e1[8:12], e2[8:12]  # four exact duplicates later in the list
    tri["p1"], tri["e1"], tri["e2"] = p1, e1, e2
    nrm = np.cross(e2, e1)
    tri["n1"] = nrm / np.linalg.norm(nrm, axis=1, keepdims=True)
    tri["material"] = np.arange(n) % 4
    mats = np.zeros(4, dtype=api.RTC_MATERIAL)
    mats["color"] = [(1, 0.2, 0.2), (0.2, 1, 0.2), (0.2, 0.2, 1), (1, 1, 0.2)]
    mats["ambient"], mats["diffuse"], mats["specular"], mats["shininess"], mats["refractive_index"] = 0.1, 0.9, 0.9, 200.0, 1.0
    groups = np.zeros(1, dtype=api.RTC_GROUP)
    groups["first"], groups["count"] = 0, n
    items = np.zeros(n, dtype=api.HREF)
    items["kind"], items["index"] = api.O_TRIANGLE, np.arange(n)
    objs = np.zeros(1, dtype=api.HREF)
    objs["kind"], objs["index"] = api.O_GROUP, 0
    lights = np.zeros(2, dtype=api.RTC_LIGHT)
    lights["position"], lights["intensity"] = [(-4, 6, -8), (5, 3, -6)], [(0.8, 0.8, 0.8), (0.4, 0.4, 0.5)]
    world = rl.RtcWorld.from_arrays(tri, mats, objs, lights, groups=groups, group_items=items)
    for cam in (rl.rtc_camera(96, 64, 1.2, (0, 0, -7), (0, 0, 0), (0, 1, 0)),          # centre column / row: d.x == 0 or d.y == 0
                rl.rtc_camera(64, 64, 0.9, (3, 2.5, -6), (0.2, 0, 0.5), (0, 1, 0))):
        gs, cs = {}, {}
        a = world.render(aa_samples=2, camera=cam, stats=gs)
        b = oracle.rtc_render(world.desc, cam, aa=2, stats=cs)
        assert np.abs(a - b).max() <= 1e-12
        for k in ("rays", "node_tests", "planar_tests", "flagged"):
            assert gs[k] == cs[k], (k, gs[k], cs[k])


@pytest.mark.parametrize("use_bvh", [False, True])
def test_coincident_spheres_resolve_ties_like_the_reference(rl, oracle, use_bvh):
    """Two spheres with identical geometry but different materials: both hits have the same t, and which one wins is decided by
    the evaluation order (sphere.rs:51-54 accepts t <= closest) — the guard ops and one-sphere LEAF visits must keep it."""
    api = rl.api
    tex = np.zeros(2, dtype=api.TEXTURE)
    tex["kind"], tex["color"] = api.TEX_SOLID, [(0.9, 0.1, 0.1), (0.1, 0.1, 0.9)]
    mats = np.zeros(3, dtype=api.MATERIAL)
    mats[0]["kind"], mats[0]["texture"] = api.MAT_LAMBERTIAN, 0
    mats[1]["kind"], mats[1]["texture"] = api.MAT_DIFFUSE_LIGHT, 1
    mats[2]["kind"], mats[2]["albedo"], mats[2]["fuzz"] = api.MAT_METAL, (0.8, 0.8, 0.8), 0.1
    sph = np.zeros(7, dtype=api.SPHERE)
    sph["center0"] = [(0, 0, -3), (0, 0, -3), (1.5, 0, -3), (1.5, 0, -3), (-1.5, 0.2, -3.5), (-1.5, 0.2, -3.5), (0, -100.5, -3)]
    sph["radius"] = [0.5, 0.5, 0.5, 0.5, 0.6, 0.6, 100.0]
    sph["material"] = [0, 1, 1, 0, 2, 0, 0]
    world = rl.World.from_spheres(sph, mats, tex, use_bvh)
    p = rl.CameraParams(aspect_ratio=2.0, image_width=96, samples_per_pixel=4, max_depth=8, vfov=50.0, lookfrom=(0, 0.3, 1), lookat=(0, 0, -3),
                        background=(0.6, 0.7, 0.9))
    _check(rl, oracle, world, p)
