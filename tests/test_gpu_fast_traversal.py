"""GPU tier: the FAST traversal of the timed sphere kernel (rtiow_wave_kernel<1024, 4, false>: ordered binary tree with reject-only
boxes, rays whose answer could depend on the visiting order re-traced by the reference's own fold — csrc/rl_fast_bvh.cpp,
rl_rtiow_wave.h).  Its frames must equal the reference-order kernels' frames BIT FOR BIT and the oracle within the tight bar, in
particular where order matters: coincident spheres (exact ties), grazing hits, axis-parallel rays, far-away origins, degenerate
spheres.  `slow_traces` (rays handed to the reference-order fold) shows that the ambiguity detection fires where it must and
almost never elsewhere."""
import numpy as np
import pytest

pytestmark = pytest.mark.gpu
TOL = 1e-4


def _frames(rl, cam, world, allow_degenerate=False):
    """(fast timed frame, reference-order timed frame, counting frame, status of the fast render)."""
    import torch
    api = rl.api
    dev = torch.device("cuda", 0)

    def timed():
        buf = torch.full((cam.c.image_height, cam.c.image_width, 3), float("nan"), dtype=torch.float64, device=dev)
        cam.render_device(world, buf.data_ptr(), stream=torch.cuda.current_stream(dev).cuda_stream)
        st = api.render_status(world, allow_degenerate=allow_degenerate)
        return buf.cpu().numpy(), st
    try:
        api.set_coop(False)  # small frames: the wave-scheduled fast kernel, not the cooperative one
        fast, st = timed()
        api.set_fast_traversal(False)
        ref_order, st0 = timed()
    finally:
        api.set_fast_traversal(True)
        api.set_coop(True)
    auto, st_auto = timed()  # the library's own choice (the cooperative kernel for small frames of sphere scenes): the same bits
    assert np.array_equal(auto, fast, equal_nan=True) and st_auto["rays"] == st["rays"] and st_auto["flagged"] == st["flagged"]
    assert st0["slow_traces"] == 0
    gs = {}
    counting = cam.render(world, stats=gs, allow_degenerate=allow_degenerate).data
    assert st["rays"] == st0["rays"] == gs["rays"] and st["flagged"] == st0["flagged"] == gs["flagged"]
    return fast, ref_order, counting, st, gs


def _same_bits(a, b):
    return np.array_equal(a.view(np.uint64), b.view(np.uint64))  # NaNs included


def _check(rl, oracle, world, p, allow_degenerate=False):
    cam = rl.Camera(p)
    fast, ref_order, counting, st, gs = _frames(rl, cam, world, allow_degenerate)
    assert _same_bits(fast, ref_order) and _same_bits(fast, counting)
    cs = {}
    cpu = oracle.rtiow_render(world.desc, cam.c, stats=cs)
    for k in ("rays", "node_tests", "sphere_tests", "rng_words", "flagged"):
        assert gs[k] == cs[k], (k, gs[k], cs[k])
    fin = np.isfinite(cpu)
    assert np.array_equal(np.isfinite(fast), fin)
    if fin.any():
        assert np.abs(fast[fin] - cpu[fin]).max() <= 1e-9 * max(1.0, np.abs(cpu[fin]).max())
    return st


def _mats(api):
    tex = np.zeros(3, dtype=api.TEXTURE)
    tex["kind"], tex["color"] = api.TEX_SOLID, [(0.9, 0.1, 0.1), (0.1, 0.1, 0.9), (0.5, 0.5, 0.5)]
    mats = np.zeros(5, dtype=api.MATERIAL)
    mats[0]["kind"], mats[0]["texture"] = api.MAT_LAMBERTIAN, 0
    mats[1]["kind"], mats[1]["texture"] = api.MAT_DIFFUSE_LIGHT, 1
    mats[2]["kind"], mats[2]["albedo"], mats[2]["fuzz"] = api.MAT_METAL, (0.8, 0.8, 0.8), 0.1
    mats[3]["kind"], mats[3]["ior"] = api.MAT_DIELECTRIC, 1.5
    mats[4]["kind"], mats[4]["texture"] = api.MAT_LAMBERTIAN, 2
    return tex, mats


def test_fast_traversal_is_the_default_for_the_baseline_scene(rl, oracle):
    world = rl.World.bouncing_spheres(1)
    p = world.params
    p.image_width, p.samples_per_pixel, p.max_depth = 160, 72, 50  # two-launch path
    st = _check(rl, oracle, world, p)
    assert st["slow_traces"] * 10_000 < st["rays"]  # the reference-order fallback is the rare path


@pytest.mark.parametrize("use_bvh", [False, True])
def test_coincident_spheres_take_the_reference_order_path(rl, oracle, use_bvh):
    """Identical geometry, different materials: equal roots, the reference's stored order decides (sphere.rs:51-54 accepts t <= closest)."""
    api = rl.api
    tex, mats = _mats(api)
    sph = np.zeros(7, dtype=api.SPHERE)
    sph["center0"] = [(0, 0, -3), (0, 0, -3), (1.5, 0, -3), (1.5, 0, -3), (-1.5, 0.2, -3.5), (-1.5, 0.2, -3.5), (0, -100.5, -3)]
    sph["radius"] = [0.5, 0.5, 0.5, 0.5, 0.6, 0.6, 100.0]
    sph["material"] = [0, 1, 1, 0, 2, 0, 4]
    world = rl.World.from_spheres(sph, mats, tex, use_bvh)
    p = rl.CameraParams(aspect_ratio=2.0, image_width=96, samples_per_pixel=4, max_depth=8, vfov=50.0, lookfrom=(0, 0.3, 1), lookat=(0, 0, -3),
                        background=(0.6, 0.7, 0.9))
    st = _check(rl, oracle, world, p)
    assert st["slow_traces"] > 500  # every ray that reaches a coincident pair


@pytest.mark.parametrize("n,use_bvh,seed", [(1, True, 1), (2, True, 2), (3, False, 3), (9, True, 4), (37, True, 5), (150, False, 6), (300, True, 7), (511, True, 8)])
def test_random_worlds_overlapping_moving_every_material(rl, oracle, n, use_bvh, seed):
    rng = np.random.default_rng(4000 + seed)
    api = rl.api
    tex, mats = _mats(api)
    sph = np.zeros(n, dtype=api.SPHERE)
    sph["center0"] = rng.uniform(-3, 3, (n, 3))
    sph["center1"] = sph["center0"] + rng.uniform(0, 0.5, (n, 3))
    sph["radius"] = rng.uniform(0.05, 0.9, n)  # heavy overlap: many candidates per ray
    sph["moving"] = rng.integers(0, 2, n)
    sph["material"] = rng.integers(0, 5, n)
    world = rl.World.from_spheres(sph, mats, tex, use_bvh)
    p = rl.CameraParams(aspect_ratio=1.5, image_width=96, samples_per_pixel=6, max_depth=12, vfov=50.0, lookfrom=(0.0, 1.0, 9.0),
                        lookat=(0.0, 0.0, 0.0), defocus_angle=1.0, focus_dist=9.0, background=(0.5, 0.6, 0.9), seed=seed)
    st = _check(rl, oracle, world, p)
    assert st["slow_traces"] * 1000 < st["rays"] + 1000


def test_too_many_spheres_or_a_far_camera_fall_back_to_the_reference_order_kernels(rl, oracle):
    api = rl.api
    tex, mats = _mats(api)
    rng = np.random.default_rng(99)
    n = 512  # one more than the fast structure's entry ids allow
    sph = np.zeros(n, dtype=api.SPHERE)
    sph["center0"], sph["radius"], sph["material"] = rng.uniform(-4, 4, (n, 3)), rng.uniform(0.05, 0.3, n), rng.integers(0, 5, n)
    world = rl.World.from_spheres(sph, mats, tex, True)
    p = rl.CameraParams(aspect_ratio=1.5, image_width=72, samples_per_pixel=4, max_depth=8, vfov=60.0, lookfrom=(0.0, 1.0, 9.0), lookat=(0, 0, 0))
    assert _check(rl, oracle, world, p)["slow_traces"] == 0
    sph = sph[:40].copy()
    world = rl.World.from_spheres(sph, mats, tex, True)
    far = rl.CameraParams(aspect_ratio=1.5, image_width=72, samples_per_pixel=4, max_depth=8, vfov=2.0, lookfrom=(0.0, 0.0, 400.0), lookat=(0, 0, 0))
    assert _check(rl, oracle, world, far)["slow_traces"] == 0  # camera beyond the frame in which the reject-only boxes are rigorous


def test_axis_parallel_rays_and_extreme_coordinates(rl, oracle):
    """Rays with a zero / denormal-scale / huge direction component or a far-away origin are outside the binary32 filter's range:
    the reference divides by zero there (aabb.rs:143-152) and its NaN / inf semantics must be reproduced, not approximated."""
    api = rl.api
    tex, mats = _mats(api)
    sph = np.zeros(5, dtype=api.SPHERE)
    sph["center0"] = [(0, 0, -4), (1.0, 0, -4), (-1.0, 0.5, -4), (0, -100.5, -4), (0, 1.0, -5)]
    sph["radius"] = [0.5, 0.5, 0.5, 100.0, 0.7]
    sph["material"] = [0, 3, 2, 4, 1]
    world = rl.World.from_spheres(sph, mats, tex, True)
    p = rl.CameraParams(aspect_ratio=1.0, image_width=33, samples_per_pixel=4, max_depth=6, vfov=40.0, lookfrom=(0, 0, 1), lookat=(0, 0, -4))
    cam = rl.Camera(p)
    # a camera whose every primary ray is exactly axis-parallel: pixel grid steps along x and y only, origin moves with the pixel
    c = cam.c
    c.pixel_du[:], c.pixel_dv[:] = (0.0, 0.0, 0.0), (0.0, 0.0, 0.0)
    c.pixel_00[:] = (0.0, 0.0, -1.0)
    c.lookfrom[:] = (0.0, 0.0, 1.0)
    c.defocus_angle = 1.0
    c.defocus_disk_u[:], c.defocus_disk_v[:] = (1.5, 0.0, 0.0), (0.0, 1.5, 0.0)  # origins spread over a disc, d = (-a, -b, -2)... not parallel
    fast, ref_order, counting, st, gs = _frames(rl, cam, world)
    assert _same_bits(fast, ref_order) and _same_bits(fast, counting)
    c.defocus_angle = 0.0  # now every primary ray is (0, 0, -2): two zero components
    fast, ref_order, counting, st, gs = _frames(rl, cam, world)
    assert _same_bits(fast, ref_order) and _same_bits(fast, counting)
    assert st["slow_traces"] >= 33 * 33 * 4  # every primary ray
    cpu = oracle.rtiow_render(world.desc, c)
    assert np.abs(fast - cpu).max() <= 1e-9 * max(1.0, np.abs(cpu).max())


def test_degenerate_sphere_scenes_do_not_use_the_fast_structure(rl, oracle):
    api = rl.api
    tex, mats = _mats(api)
    sph = np.zeros(3, dtype=api.SPHERE)
    sph["center0"] = [(0, 0, -1), (0.6, 0, -1), (0, -100.5, -1)]
    sph["radius"] = [0.5, 0.0, 100.0]  # radius 0: (p - c) / r is not finite -> from_normalized would panic (vec3.rs:219)
    sph["material"] = [0, 0, 4]
    world = rl.World.from_spheres(sph, mats, tex, False)
    p = rl.CameraParams(aspect_ratio=1.0, image_width=32, samples_per_pixel=2, max_depth=4, lookfrom=(0, 0, 1), lookat=(0, 0, -1))
    assert _check(rl, oracle, world, p, allow_degenerate=True)["slow_traces"] == 0
    sph["center0"][:, 0] += 1e12  # unit-sized spheres at x = 1e12: thousands of flagged hits (see test_gpu_timed_kernels)
    sph["radius"] = [0.5, 0.4, 100.0]
    world = rl.World.from_spheres(sph, mats, tex, True)
    p = rl.CameraParams(aspect_ratio=1.0, image_width=32, samples_per_pixel=4, max_depth=4, lookfrom=(1e12, 0, 1), lookat=(1e12, 0, -1))
    cam = rl.Camera(p)
    fast, ref_order, counting, st, gs = _frames(rl, cam, world, allow_degenerate=True)
    assert _same_bits(fast, ref_order) and _same_bits(fast, counting) and gs["flagged"] > 100


@pytest.mark.parametrize("scene", ["horizon", "far_camera"])
def test_guarded_and_unguarded_ops_render_the_same_bits(rl, scene):
    """The counting kernel's guard ops (variant 1027: a sphere's own padded box may skip its Sphere::hit) against the same kernel without
    them (1025), on rays that graze spheres and on origins far from tiny spheres — where a wrong reject would show."""
    api = rl.api
    tex, mats = _mats(api)
    if scene == "horizon":
        sph = np.zeros(4, dtype=api.SPHERE)
        sph["center0"] = [(0, -1000, 0), (0, 0.5, -30), (3, 0.2, -20), (-4, 1.0, -40)]
        sph["radius"] = [1000.0, 0.5, 0.2, 1.0]
        sph["material"] = [4, 3, 2, 0]
        p = rl.CameraParams(aspect_ratio=4.0, image_width=256, samples_per_pixel=8, max_depth=10, vfov=3.0, lookfrom=(0, 0.02, 10), lookat=(0, -0.45, -90), seed=5)
    else:  # 0.01-radius spheres seen from 400 units: |oc| / r = 4e4
        rng = np.random.default_rng(3)
        sph = np.zeros(60, dtype=api.SPHERE)
        sph["center0"], sph["radius"], sph["material"] = rng.uniform(-0.5, 0.5, (60, 3)), 0.01, rng.integers(0, 5, 60)
        p = rl.CameraParams(aspect_ratio=1.0, image_width=128, samples_per_pixel=8, max_depth=6, vfov=0.2, lookfrom=(0, 0, 400.0), lookat=(0, 0, 0), seed=6)
    world = rl.World.from_spheres(sph, mats, tex, True)
    cam = rl.Camera(p)
    frames = {}
    try:
        for v in (1027, 1025):
            api.set_rtiow_variant(v)
            st = {}
            frames[v] = (cam.render(world, stats=st).data, st)
    finally:
        api.set_rtiow_variant(0)
    assert _same_bits(frames[1027][0], frames[1025][0])
    for k in ("rays", "node_tests", "sphere_tests", "rng_words", "flagged"):
        assert frames[1027][1][k] == frames[1025][1][k], k


def test_grazing_rays_over_a_sphere_horizon(rl, oracle):
    """A camera sitting on a huge sphere looking along its surface: most primary rays graze the ground sphere or pass just above it."""
    api = rl.api
    tex, mats = _mats(api)
    sph = np.zeros(4, dtype=api.SPHERE)
    sph["center0"] = [(0, -1000, 0), (0, 0.5, -30), (3, 0.2, -20), (-4, 1.0, -40)]
    sph["radius"] = [1000.0, 0.5, 0.2, 1.0]
    sph["material"] = [4, 3, 2, 0]
    world = rl.World.from_spheres(sph, mats, tex, True)
    p = rl.CameraParams(aspect_ratio=4.0, image_width=256, samples_per_pixel=8, max_depth=10, vfov=3.0, lookfrom=(0, 0.02, 10), lookat=(0, -0.45, -90),
                        background=(0.6, 0.7, 0.9), seed=5)
    _check(rl, oracle, world, p)


# ------------------------------------------------------------------------------------------------ general scenes (rl_rtiow_fastgen.h)
def _spot_texture():
    import os
    from PIL import Image
    root = os.path.dirname(os.path.abspath(__file__))
    return np.asarray(Image.open(os.path.join(root, "golden", "spot_texture.png")).convert("RGB"))


def _general_scenes(rl, golden):
    tex = _spot_texture()
    lin = (tex.astype(np.float32) / 255.0) ** 2.2

    def composed(b):  # every primitive kind, an instance chain of scale -> rotate_x -> rotate_z -> translate around a BVH
        red = b.lambertian(b.solid((0.8, 0.2, 0.2)))
        chk = b.lambertian(b.checker(0.5, b.solid((0.1, 0.1, 0.1)), b.solid((0.9, 0.9, 0.9))))
        mirror = b.metal((0.8, 0.8, 0.8), 0.05)
        glass = b.dielectric(1.5)
        light = b.diffuse_light(b.solid((4.0, 4.0, 4.0)))
        objs = [b.sphere((0, 0, -1), 0.5, red), b.sphere((1.2, 0, -1.5), 0.5, glass), b.sphere((-1.2, 0.1, -1.2), 0.4, mirror, center2=(-1.2, 0.4, -1.2)),
                b.quad((-3, -0.5, -4), (6, 0, 0), (0, 0, 5), chk), b.triangle((-1, 0.8, -2), (2, 0, 0), (0, 1.5, 0), light)]
        tri = b.triangle_from_model([[0, 0, 0], [1, 0, 0], [0, 1, 0]], red, uvs=[[0, 0], [1, 0], [0, 1]], normals=[[0, 0, 1], [0.2, 0, 1], [0, 0.2, 1]])
        inner = b.bvh([b.sphere((0, 0, 0), 0.3, red), tri, b.quad((0, 0, 0.2), (0.5, 0, 0), (0, 0.5, 0), glass)])
        inst = b.translate(b.rotate_z(b.rotate_x(b.scale(inner, 1.5), 25.0), -40.0), (0.3, 0.9, -1.8))
        return b.bvh(objs + [inst])

    def cornell(b):
        white = b.lambertian(b.solid((0.73, 0.73, 0.73)))
        green = b.lambertian(b.solid((0.12, 0.45, 0.15)))
        img = b.lambertian(b.image(lin[::32, ::32].copy()))
        light = b.diffuse_light(b.solid((15, 15, 15)))
        qs = [b.quad((555, 0, 0), (0, 555, 0), (0, 0, 555), green), b.quad((0, 0, 0), (0, 555, 0), (0, 0, 555), img),
              b.quad((343, 554, 332), (-130, 0, 0), (0, 0, -105), light), b.quad((0, 0, 0), (555, 0, 0), (0, 0, 555), white),
              b.quad((555, 555, 555), (-555, 0, 0), (0, 0, -555), white), b.quad((0, 0, 555), (555, 0, 0), (0, 555, 0), white),
              b.translate(b.rotate_y(b.sphere((0, 0, 0), 90.0, b.dielectric(1.5)), 15.0), (190, 90, 190)),
              b.quad((130, 0, 65), (165, 0, 0), (0, 165, 0), white), b.quad((130, 0, 65), (165, 0, 0), (0, 165, 0), green)]  # two coincident quads
        return b.list(qs)

    def instanced_twice(b):  # one mesh placed twice: two occurrences of every triangle
        mat = b.lambertian(b.solid((0.6, 0.6, 0.7)))
        mesh = b.bvh([b.triangle((0, 0, 0), (1, 0, 0), (0, 1, 0), mat), b.triangle((1, 0, 0), (0, 1, 0), (0, 0, 1), mat),
                      b.quad((0, 0, 0), (0, 0, 1), (0, 1, 0), mat), b.sphere((0.4, 0.4, 0.4), 0.25, b.metal((0.9, 0.8, 0.7), 0.0))])
        return b.bvh([b.translate(mesh, (-1.2, 0, -3)), b.translate(b.rotate_y(mesh, 70.0), (0.6, -0.2, -3.5)),
                      b.sphere((0, -100.5, -3), 100, b.lambertian(b.checker(0.7, b.solid((0.2, 0.3, 0.1)), b.solid((0.9, 0.9, 0.9))))),
                      b.quad((-2, 1.8, -5), (4, 0, 0), (0, 0, 3), b.diffuse_light(b.solid((3, 3, 3))))])

    cam1 = rl.CameraParams(aspect_ratio=1.5, image_width=120, samples_per_pixel=6, max_depth=12, vfov=60.0, lookfrom=(0.2, 0.6, 2.5), lookat=(0.0, 0.2, -1.0),
                           defocus_angle=0.5, focus_dist=3.0, background=(0.3, 0.4, 0.6), seed=3)
    cam2 = rl.CameraParams(aspect_ratio=1.0, image_width=64, samples_per_pixel=8, max_depth=20, vfov=40.0, lookfrom=(278, 278, -800), lookat=(278, 278, 0), background=(0, 0, 0))
    cam3 = rl.CameraParams(aspect_ratio=1.5, image_width=96, samples_per_pixel=66, max_depth=10, vfov=50.0, lookfrom=(0, 0.5, 1.0), lookat=(0, 0, -3), background=(0.6, 0.7, 0.9), seed=9)
    cow = rl.World.cow_scene(golden("spot_triangulated.obj.gz"), tex)
    pc = cow.params
    pc.aspect_ratio, pc.image_width, pc.samples_per_pixel = 16.0 / 9.0, 128, 8
    stress = rl.World.stress_scene(60, 1, golden("spot_triangulated.obj.gz"), tex)
    ps = stress.params
    ps.image_width, ps.samples_per_pixel, ps.max_depth = 128, 4, 50
    earth = rl.World.earth_scene(tex)
    pe = earth.params
    pe.image_width, pe.samples_per_pixel, pe.max_depth = 96, 6, 20
    return [("composed", rl.World.build(composed), cam1), ("cornell", rl.World.build(cornell), cam2), ("instanced_twice", rl.World.build(instanced_twice), cam3),
            ("cow", cow, pc), ("stress", stress, ps), ("earth", earth, pe), ("perlin", rl.World.perlin_spheres(), None), ("simple_light", rl.World.simple_light(), None)]


def test_general_scenes_fast_traversal_equals_reference_order_and_oracle(rl, oracle, golden):
    """Planars, Translate / Transform chains, image / noise textures, a mesh instanced twice, coincident quads, the cfg-4 and cfg-5
    generators: the fast world-space traversal renders the reference-order kernels' frames bit for bit."""
    report = {}
    for name, world, p in _general_scenes(rl, golden):
        if p is None:
            p = world.params
            p.image_width, p.samples_per_pixel, p.max_depth = 96, 6, 20
        cam = rl.Camera(p)
        fast, ref_order, counting, st, gs = _frames(rl, cam, world)
        assert _same_bits(fast, ref_order) and _same_bits(fast, counting), name
        cs = {}
        cpu = oracle.rtiow_render(world.desc, cam.c, stats=cs)
        for k in ("rays", "node_tests", "sphere_tests", "planar_tests", "instance_enters", "rng_words", "flagged"):
            assert gs[k] == cs[k], (name, k, gs[k], cs[k])
        assert np.abs(fast - cpu).max() <= 1e-9 * max(1.0, np.abs(cpu).max()), name
        report[name] = (st["slow_traces"], st["rays"])
    print(report)
    assert report["cornell"][0] > 0                       # the coincident quads tie
    assert report["cow"][0] * 1000 < report["cow"][1]     # and the BASELINE scenes almost never fall back
    assert report["stress"][0] * 200 < report["stress"][1]


def test_general_kernel_pending_list_overflow_falls_back_to_the_reference_order(rl, oracle):
    """600 concentric glass / diffuse shells plus a quad (so the scene runs the general fast kernel): the surface-area heuristic peels
    the shells one by one into a tree as deep as its budget, a ray towards the centre hits every box of it, and the list of pending
    children outgrows the kernel's 20-entry LDS stack — such rays must be re-traced in the reference's order, not truncated."""

    def shells(b):
        glass, red = b.dielectric(1.3), b.lambertian(b.solid((0.7, 0.3, 0.3)))
        objs = [b.sphere((0.0, 0.0, -3.0), 0.02 * 1.012 ** k, glass if k % 7 else red) for k in range(600)]
        objs.append(b.quad((-4, -2.6, -7), (8, 0, 0), (0, 0, 8), b.lambertian(b.checker(0.8, b.solid((0.2, 0.2, 0.2)), b.solid((0.9, 0.9, 0.9))))))
        return b.bvh(objs)

    world = rl.World.build(shells)
    p = rl.CameraParams(aspect_ratio=1.0, image_width=48, samples_per_pixel=4, max_depth=30, vfov=50.0, lookfrom=(0, 0.3, 2.0), lookat=(0, 0, -3),
                        background=(0.6, 0.7, 0.9), seed=4)
    cam = rl.Camera(p)
    fast, ref_order, counting, st, gs = _frames(rl, cam, world)
    assert _same_bits(fast, ref_order) and _same_bits(fast, counting)
    cs = {}
    cpu = oracle.rtiow_render(world.desc, cam.c, stats=cs)
    for k in ("rays", "node_tests", "sphere_tests", "planar_tests", "rng_words", "flagged"):
        assert gs[k] == cs[k], (k, gs[k], cs[k])
    assert np.abs(fast - cpu).max() <= 1e-9 * max(1.0, np.abs(cpu).max())
    print("slow traces", st["slow_traces"], "of", st["rays"])
    assert st["slow_traces"] > 0


def test_vanishing_interpolated_normals_keep_the_reference_order(rl, oracle):
    """triangle.rs:78-83 normalises n2 a + n3 b + n1 (1 - a - b).  Vertex normals that merely point apart (the teapot's lid: pairwise
    dot products <= 0, shortest interpolated normal 0.43) are fine for the fast traversal; a triangle whose interpolated normal passes
    through zero is a reference panic site whose count depends on the visiting order, so such a scene must stay on the reference-order
    kernels — either way the frames and the panic-site count equal the oracle's."""

    def scene(vanishing):
        def build(b):
            mat = b.lambertian(b.solid((0.7, 0.6, 0.5)))
            apart = b.triangle_from_model([[-1, 0, -3], [1, 0, -3], [0, 1.5, -3]], mat, normals=[[-0.8, 0, 0.6], [0.8, 0, 0.6], [0, 0.9, 0.45]])
            n3 = [0, 0, -1] if vanishing else [0, 0.6, 0.8]
            other = b.triangle_from_model([[-1, -1.6, -2.5], [1, -1.6, -2.5], [0, -0.2, -2.5]], mat, normals=[[0, 0, 1], [0.6, 0, 0.8], n3])
            return b.bvh([apart, other, b.quad((-3, -1.7, -5), (6, 0, 0), (0, 0, 5), mat), b.sphere((0, 0.3, -1.5), 0.2, b.dielectric(1.5))])
        return rl.World.build(build)

    p = rl.CameraParams(aspect_ratio=1.0, image_width=64, samples_per_pixel=8, max_depth=8, vfov=70.0, lookfrom=(0, 0, 1), lookat=(0, 0, -3),
                        background=(0.5, 0.6, 0.8), seed=2)
    cam = rl.Camera(p)
    for vanishing in (False, True):
        world = scene(vanishing)
        fast, ref_order, counting, st, gs = _frames(rl, cam, world, allow_degenerate=True)
        assert _same_bits(fast, ref_order) and _same_bits(fast, counting), vanishing
        cs = {}
        cpu = oracle.rtiow_render(world.desc, cam.c, stats=cs)
        for k in ("rays", "node_tests", "planar_tests", "rng_words", "flagged"):
            assert gs[k] == cs[k], (vanishing, k, gs[k], cs[k])
        assert np.nanmax(np.abs(fast - cpu)) <= 1e-9 * max(1.0, np.nanmax(np.abs(cpu))), vanishing
