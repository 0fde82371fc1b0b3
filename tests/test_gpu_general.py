"""GPU tier: the all-primitives RTIOW kernel (planes / quads / triangles, Translate / Transform instances,
Image textures, scene read from HBM) against the CPU oracle.  Same bars as test_gpu_parity.py:
control-flow counters EXACT, colours within 1e-4 (north_star) and in practice ~1e-13."""
import gzip
import os

import numpy as np
import pytest

pytestmark = pytest.mark.gpu
TOL = 1e-4
COUNTERS = ("rays", "node_tests", "sphere_tests", "planar_tests", "instance_enters", "rng_words", "flagged")


def _parity(rl, oracle, world, p, tight=1e-9):
    cam = rl.Camera(p)
    gs, cs = {}, {}
    gpu = cam.render(world, stats=gs).data
    cpu = oracle.rtiow_render(world.desc, cam.c, stats=cs)
    for k in COUNTERS:
        assert gs[k] == cs[k], (k, gs[k], cs[k])
    assert np.abs(gpu - cpu).max() / p.samples_per_pixel <= TOL
    assert np.abs(gpu - cpu).max() <= tight * max(1.0, np.abs(cpu).max())
    return gs


def _spot_texture():
    from PIL import Image
    root = os.path.dirname(os.path.abspath(__file__))
    return np.asarray(Image.open(os.path.join(root, "golden", "spot_texture.png")).convert("RGB"))


def test_cow_scene_cfg4_reduced(rl, oracle, golden):
    # BASELINE cfg 4 scene (examples/cow.rs: 5856 textured triangles under scale->rotate_y->translate, Cornell quads,
    # light quad) at 16:9, reduced to 160x90, 8 spp, depth 40
    world = rl.World.cow_scene(golden("spot_triangulated.obj.gz"), _spot_texture())
    p = world.params
    p.aspect_ratio, p.image_width, p.samples_per_pixel = 16.0 / 9.0, 160, 8
    gs = _parity(rl, oracle, world, p)
    assert gs["planar_tests"] > 0 and gs["instance_enters"] > 0


def test_perlin_spheres_and_simple_light_vs_oracle(rl, oracle):
    """examples/perlin_spheres.rs and simple_light.rs: Noise{Perlin} textures (perlin.rs, texture.rs:84-94), a quad light and
    a sphere light.  The Perlin tables come from the host mirror of Perlin::new (unpinned); noise()/turb() run on the device."""
    for world in (rl.World.perlin_spheres(), rl.World.simple_light()):
        p = world.params
        p.image_width, p.samples_per_pixel, p.max_depth = 96, 6, 20
        _parity(rl, oracle, world, p)


def test_noise_texture_through_the_scene_builder(rl, oracle):
    def scene(b):
        n1 = b.lambertian(b.noise(4.0, 7))
        n2 = b.diffuse_light(b.checker(0.5, b.noise(1.5, 8), b.solid((0.2, 0.9, 0.3))))
        objs = [b.sphere((0, -1000, 0), 1000, n1), b.sphere((0, 2, 0), 2, n2), b.sphere((3, 1, 2), 1, b.dielectric(1.5)),
                b.quad((-4, 0, -3), (3, 0, 0), (0, 3, 0), n1)]
        return b.bvh(objs)
    world = rl.World.build(scene)
    p = rl.CameraParams(aspect_ratio=1.5, image_width=96, samples_per_pixel=6, max_depth=12, vfov=30.0, lookfrom=(13, 3, 5), lookat=(0, 1, 0))
    _parity(rl, oracle, world, p)


def test_earth_scene_image_texture_on_a_sphere(rl, oracle, golden):
    """examples/earth.rs: Lambertian{Image} on a sphere needs get_sphere_uv (sphere.rs:91-99).  earthmap.jpg is not decoded
    here; the spot texture stands in for it (same code path)."""
    tex = _spot_texture()
    world = rl.World.earth_scene(tex)
    p = world.params
    p.image_width, p.samples_per_pixel, p.max_depth = 96, 6, 20
    _parity(rl, oracle, world, p)
    # and inside a rotated / translated instance next to a moving textured sphere (uv is taken in object space)
    lin = (tex.astype(np.float32) / 255.0) ** 2.2

    def scene(b):
        m = b.lambertian(b.image(lin))
        s1 = b.translate(b.rotate_y(b.sphere((0, 0, 0), 1.5, m), 30.0), (1.0, 0.5, 0.0))
        s2 = b.sphere((-3, 0, 0), 1.0, m, center2=(-3, 0.5, 0))
        return b.bvh([s1, s2, b.sphere((0, -101.5, 0), 100, b.lambertian(b.solid((0.5, 0.5, 0.5))))])
    world = rl.World.build(scene)
    p = rl.CameraParams(aspect_ratio=1.5, image_width=96, samples_per_pixel=6, max_depth=10, vfov=30.0, lookfrom=(0, 2, 12), lookat=(0, 0, 0))
    _parity(rl, oracle, world, p)


def test_composed_scene_all_primitive_kinds(rl, oracle):
    def scene(b):
        red = b.lambertian(b.solid((0.8, 0.2, 0.2)))
        chk = b.lambertian(b.checker(0.5, b.solid((0.1, 0.1, 0.1)), b.solid((0.9, 0.9, 0.9))))
        mirror = b.metal((0.8, 0.8, 0.8), 0.05)
        glass = b.dielectric(1.5)
        light = b.diffuse_light(b.solid((4.0, 4.0, 4.0)))
        objs = [b.sphere((0, 0, -1), 0.5, red), b.sphere((1.2, 0, -1.5), 0.5, glass), b.sphere((-1.2, 0.1, -1.2), 0.4, mirror, center2=(-1.2, 0.4, -1.2)),
                b.quad((-3, -0.5, -4), (6, 0, 0), (0, 0, 5), chk), b.triangle((-1, 0.8, -2), (2, 0, 0), (0, 1.5, 0), light),
                b.plane((0, 0, -6), (1, 0, 0), (0, 1, 0), mirror)]
        tri = b.triangle_from_model([[0, 0, 0], [1, 0, 0], [0, 1, 0]], red, uvs=[[0, 0], [1, 0], [0, 1]], normals=[[0, 0, 1], [0.2, 0, 1], [0, 0.2, 1]])
        inner = b.bvh([b.sphere((0, 0, 0), 0.3, red), tri, b.quad((0, 0, 0.2), (0.5, 0, 0), (0, 0.5, 0), glass)])
        inst = b.translate(b.rotate_z(b.rotate_x(b.scale(inner, 1.5), 25.0), -40.0), (0.3, 0.9, -1.8))
        return b.bvh(objs[:5] + [inst]) if False else b.list([b.bvh(objs[:5] + [inst]), objs[5]])
    world = rl.World.build(scene)
    p = rl.CameraParams(aspect_ratio=1.5, image_width=120, samples_per_pixel=6, max_depth=12, vfov=60.0, lookfrom=(0.2, 0.6, 2.5),
                        lookat=(0.0, 0.2, -1.0), defocus_angle=0.5, focus_dist=3.0, background=(0.3, 0.4, 0.6), seed=3)
    gs = _parity(rl, oracle, world, p)
    assert gs["planar_tests"] > 0 and gs["instance_enters"] > 0 and gs["sphere_tests"] > 0


def test_cornell_quads_and_image_textured_quad(rl, oracle):
    tex = np.linspace(0, 1, 16 * 8 * 3, dtype=np.float32).reshape(8, 16, 3)

    def scene(b):
        white = b.lambertian(b.solid((0.73, 0.73, 0.73)))
        green = b.lambertian(b.solid((0.12, 0.45, 0.15)))
        img = b.lambertian(b.image(tex))
        light = b.diffuse_light(b.solid((15, 15, 15)))
        qs = [b.quad((555, 0, 0), (0, 555, 0), (0, 0, 555), green), b.quad((0, 0, 0), (0, 555, 0), (0, 0, 555), img),
              b.quad((343, 554, 332), (-130, 0, 0), (0, 0, -105), light), b.quad((0, 0, 0), (555, 0, 0), (0, 0, 555), white),
              b.quad((555, 555, 555), (-555, 0, 0), (0, 0, -555), white), b.quad((0, 0, 555), (555, 0, 0), (0, 555, 0), white)]
        return b.bvh(qs)
    world = rl.World.build(scene)
    p = rl.CameraParams(aspect_ratio=1.0, image_width=64, samples_per_pixel=8, max_depth=20, vfov=40.0, lookfrom=(278, 278, -800),
                        lookat=(278, 278, 0), background=(0, 0, 0))
    _parity(rl, oracle, world, p)


def test_general_kernel_equals_wave_kernel_on_spheres(rl):
    """Every RTIOW kernel variant of the product library renders the same bits and counts the same calls."""
    world = rl.World.bouncing_spheres(1)
    p = world.params
    p.image_width, p.samples_per_pixel, p.max_depth = 120, 4, 50
    cam = rl.Camera(p)
    st0 = {}
    a = cam.render(world, stats=st0).data
    try:
        # 2 = nested-loop all-primitives kernel, 4 = wave-scheduled all-primitives kernel, 1027 = guarded compact ops in LDS (the
        # counting renders' default), 1025 = 1024 lanes with 64-byte linked ops (no guards), 1024 = scene through L2
        for v in (2, 4, 1027, 1025, 1024):
            rl.api.set_rtiow_variant(v)
            sv = {}
            img = cam.render(world, stats=sv).data
            assert np.array_equal(a, img), v
            for k in ("rays", "node_tests", "sphere_tests", "rng_words", "flagged"):
                assert sv[k] == st0[k], (v, k)
    finally:
        rl.api.set_rtiow_variant(0)


def test_stress_scene_cfg5_reduced(rl, oracle, golden):
    # BASELINE configs[4] generator (grid of spheres + ground + subdivided spot mesh under an instance chain), reduced:
    # 60x60 spheres, one subdivision (23,424 triangles), 128x72, 4 spp, depth 50
    world = rl.World.stress_scene(60, 1, golden("spot_triangulated.obj.gz"), _spot_texture())
    p = world.params
    p.image_width, p.samples_per_pixel, p.max_depth = 128, 4, 50
    gs = _parity(rl, oracle, world, p)
    assert gs["planar_tests"] > 0 and gs["sphere_tests"] > 0 and gs["instance_enters"] > 0
