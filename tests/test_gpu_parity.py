"""GPU tier (-m gpu): the HIP path, called through the C ABI, against the CPU oracle and the
reference's golden images.

Bars (BASELINE.json north_star): pixel RGB within 1e-4 per channel of the seeded CPU reference.
Control flow is integer work and must be EXACT: ray / AABB-test / sphere-test / ChaCha-word counters
of the GPU run must equal the oracle's, which proves every branch of every path went the same way.
Colour sums may differ by reassociation only (throughput-form bounce loop): asserted <= 1e-9 relative,
far inside the 1e-4 bar.
"""
import numpy as np
import pytest

pytestmark = pytest.mark.gpu

TOL = 1e-4  # north_star: per-channel tolerance on pixel RGB (means)
COUNTERS = ("rays", "node_tests", "sphere_tests", "rng_words", "flagged")


def _assert_rtiow_parity(gpu_sums, cpu_sums, spp, gs, cs):
    for k in COUNTERS:
        assert gs[k] == cs[k], (k, gs[k], cs[k])
    err = np.abs(gpu_sums - cpu_sums).max() / spp
    assert err <= TOL
    scale = max(1.0, np.abs(cpu_sums).max())
    assert np.abs(gpu_sums - cpu_sums).max() <= 1e-9 * scale, np.abs(gpu_sums - cpu_sums).max()


def test_rtiow_golden_scene_ppm_byte_exact_on_gpu(rl, oracle, golden):
    world = rl.World.golden_test_scene()
    cam = rl.Camera(world.params)
    gs, cs = {}, {}
    canvas = cam.render(world, stats=gs)
    assert rl.output_ppm(canvas).encode() == golden("test.ppm.gz")
    cpu = oracle.rtiow_render(world.desc, cam.c, stats=cs)
    _assert_rtiow_parity(canvas.data, cpu, world.params.samples_per_pixel, gs, cs)


def test_rtiow_bouncing_spheres_cfg1_reduced_vs_oracle(rl, oracle):
    # BASELINE cfg 1 scene and camera (400x225, depth 50) at 8 spp so the oracle finishes in seconds
    world = rl.World.bouncing_spheres(1)
    p = world.params
    p.max_depth, p.samples_per_pixel = 50, 8
    cam = rl.Camera(p)
    assert (cam.c.image_width, cam.c.image_height) == (400, 225)
    gs, cs = {}, {}
    canvas = cam.render(world, stats=gs)
    cpu = oracle.rtiow_render(world.desc, cam.c, stats=cs)
    _assert_rtiow_parity(canvas.data, cpu, p.samples_per_pixel, gs, cs)
    assert gs["rays"] > 400 * 225 * 8


def test_rtiow_row_shards_equal_full_render(rl):
    world = rl.World.bouncing_spheres(1)
    p = world.params
    p.image_width, p.max_depth, p.samples_per_pixel = 160, 50, 4
    cam = rl.Camera(p)
    full = cam.render(world).data
    for G in (2, 3, 8):
        for g in range(G):
            part = cam.render_rows(world, g, G)
            assert np.array_equal(part, full[g::G]), (G, g)


def test_rtiow_determinism_and_checkpointing(rl):
    # mirrors tests/ray_tracing_one_weekend.rs:97-162
    world = rl.World.golden_test_scene()
    p = world.params
    p.samples_per_pixel, p.image_width = 5, 100
    cam = rl.Camera(p)
    r1, r2 = cam.render(world), cam.render(world)
    assert np.array_equal(r1.pixel_data(), r2.pixel_data())
    c1 = cam.render_from_checkpoint(world, r1)
    c2 = cam.render_from_checkpoint(world, r1)
    assert not np.array_equal(r1.pixel_data(), c1.pixel_data())
    assert np.array_equal(c1.pixel_data(), c2.pixel_data())
    assert c1.samples == 10
    p10 = rl.CameraParams(**{**p.__dict__, "samples_per_pixel": 10})
    hq = rl.Camera(p10).render(world)
    assert not np.array_equal(c1.pixel_data(), hq.pixel_data())


def test_rtiow_checkpoint_matches_oracle(rl, oracle):
    world = rl.World.golden_test_scene()
    p = world.params
    p.samples_per_pixel, p.image_width = 5, 100
    cam = rl.Camera(p)
    gpu = cam.render_rows(world, 0, 1, first_sample=5)
    cpu = oracle.rtiow_render(world.desc, cam.c, first_sample=5)
    assert np.abs(gpu - cpu).max() / 5 <= TOL and np.abs(gpu - cpu).max() <= 1e-9


@pytest.mark.parametrize("n,use_bvh", [(1, False), (1, True), (2, True), (3, True), (7, False), (37, True), (300, True)])
def test_rtiow_random_sphere_worlds_vs_oracle(rl, oracle, n, use_bvh):
    rng = np.random.default_rng(1234 + n)
    api = rl.api
    tex = np.zeros(3, dtype=api.TEXTURE)
    tex[0]["kind"], tex[0]["color"] = api.TEX_SOLID, (0.2, 0.3, 0.1)
    tex[1]["kind"], tex[1]["color"] = api.TEX_SOLID, (0.9, 0.9, 0.9)
    tex[2]["kind"], tex[2]["even"], tex[2]["odd"], tex[2]["inv_scale"] = api.TEX_CHECKER, 0, 1, 1.0 / 0.32
    mats = np.zeros(5, dtype=api.MATERIAL)
    mats[0]["kind"], mats[0]["texture"] = api.MAT_LAMBERTIAN, 2
    mats[1]["kind"], mats[1]["albedo"], mats[1]["fuzz"] = api.MAT_METAL, (0.8, 0.6, 0.2), 0.3
    mats[2]["kind"], mats[2]["ior"] = api.MAT_DIELECTRIC, 1.5
    mats[3]["kind"], mats[3]["texture"] = api.MAT_DIFFUSE_LIGHT, 1
    mats[4]["kind"], mats[4]["texture"] = api.MAT_LAMBERTIAN, 0
    sph = np.zeros(n, dtype=api.SPHERE)
    sph["center0"] = rng.uniform(-3, 3, (n, 3))
    sph["center1"] = sph["center0"] + rng.uniform(0, 0.5, (n, 3))
    sph["radius"] = rng.uniform(0.1, 0.8, n)
    sph["moving"] = rng.integers(0, 2, n)
    sph["material"] = rng.integers(0, 5, n)
    world = rl.World.from_spheres(sph, mats, tex, use_bvh)
    p = rl.CameraParams(aspect_ratio=1.5, image_width=96, samples_per_pixel=6, max_depth=12, vfov=50.0, lookfrom=(0.0, 1.0, 9.0),
                        lookat=(0.0, 0.0, 0.0), defocus_angle=1.0, focus_dist=9.0, background=(0.5, 0.6, 0.9), seed=7)
    cam = rl.Camera(p)
    gs, cs = {}, {}
    canvas = cam.render(world, stats=gs)
    cpu = oracle.rtiow_render(world.desc, cam.c, stats=cs)
    _assert_rtiow_parity(canvas.data, cpu, p.samples_per_pixel, gs, cs)


def test_rtc_golden_obj_scene_ppm_byte_exact_on_gpu(rl, oracle, golden):
    world = rl.RtcWorld.test_obj_scene(golden("teapot-low.obj"), 300, 200)
    gs, cs = {}, {}
    img = world.render(1, stats=gs)
    assert rl.canvas_ppm(img).encode() == golden("test_obj_scene.ppm.gz")
    cpu = oracle.rtc_render(world.desc, world.camera, aa=1, stats=cs)
    for k in ("rays", "node_tests", "planar_tests", "instance_enters", "flagged"):
        assert gs[k] == cs[k], (k, gs[k], cs[k])
    assert np.abs(img - cpu).max() <= TOL
    assert np.abs(img - cpu).max() <= 1e-12  # device pow() vs glibc pow(): <= a few ulps of O(1) colours


def test_rtc_teapot_antialiased_and_sharded(rl, oracle, golden):
    world = rl.RtcWorld.test_obj_scene(golden("teapot-low.obj"), 240, 135)
    img = world.render(3)
    cpu = oracle.rtc_render(world.desc, world.camera, aa=3)
    assert np.abs(img - cpu).max() <= 1e-12
    for g in range(4):
        assert np.array_equal(world.render(3, row_first=g, row_step=4), img[g::4])
    # the grid is the number of workgroups resident at once (occupancy API); any other count strides over the same pixels: same frame, same counters
    L = rl.api.render_lib()
    L.rl_debug_set_rtc_blocks.argtypes = [rl.api.C.c_int]
    ref_stats = {}
    world.render(3, stats=ref_stats)
    mirror = rl.RtcWorld.test_mirror_scene(120, 80)  # the full World::color_at kernel
    mref = mirror.render(1)
    try:
        for per_cu in (1, 3, 8, 64):
            L.rl_debug_set_rtc_blocks(per_cu)
            st = {}
            assert np.array_equal(world.render(3, stats=st), img), per_cu
            assert all(st[k] == ref_stats[k] for k in ("rays", "node_tests", "planar_tests", "instance_enters", "flagged")), per_cu
            assert np.array_equal(mirror.render(1), mref), per_cu
    finally:
        L.rl_debug_set_rtc_blocks(0)


def test_rtiow_cost_sorted_two_phase_render_is_bit_identical(rl, oracle):
    # spp >= 64 takes the LPT path (8-sample launch, tiles sorted by cost, resumed launch): same bits as one launch
    world = rl.World.bouncing_spheres(1)
    p = world.params
    p.image_width, p.samples_per_pixel, p.max_depth = 100, 64, 50
    cam = rl.Camera(p)
    L = rl.api.render_lib()
    gs = {}
    a = cam.render(world, stats=gs).data
    try:
        L.rl_debug_set_lpt(0)
        gs0 = {}
        b = cam.render(world, stats=gs0).data
    finally:
        L.rl_debug_set_lpt(1)
    assert np.array_equal(a, b)
    for k in COUNTERS:
        assert gs[k] == gs0[k], k
    cs = {}
    cpu = oracle.rtiow_render(world.desc, cam.c, stats=cs)
    _assert_rtiow_parity(a, cpu, p.samples_per_pixel, gs, cs)


def test_device_output_stage_matches_golden_bytes(rl, golden):
    # pixel_data / linear_to_srgb / floor(v*255.999) (RTIOW) and round(c*255) (RTC) on the GPU
    world = rl.World.golden_test_scene()
    rgb8 = rl.Camera(world.params).render_rgb8(world)
    want = np.array(golden("test.ppm.gz").split()[4:], dtype=np.uint8).reshape(168, 300, 3)
    assert np.array_equal(rgb8, want)
    rw = rl.RtcWorld.test_obj_scene(golden("teapot-low.obj"), 300, 200)
    want = np.array(golden("test_obj_scene.ppm.gz").split()[4:], dtype=np.uint8).reshape(200, 300, 3)
    assert np.array_equal(rw.render_rgb8(1), want)
