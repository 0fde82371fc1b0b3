"""GPU tier: RTC full World::color_at kernel (shapes, CSG, patterns, reflection / refraction) vs the oracle and the
reference goldens test_csg_scene.ppm / test_mirror_scene.ppm."""
import numpy as np
import pytest

pytestmark = pytest.mark.gpu
TOL = 1e-4
COUNTERS = ("rays", "node_tests", "sphere_tests", "planar_tests", "instance_enters", "flagged")


def _ppm_array(ppm, w=300, h=200):
    return np.array(ppm.split()[4:], dtype=int).reshape(h, w, 3)


def test_rtc_csg_scene_golden_on_gpu(rl, oracle, golden):
    world = rl.RtcWorld.test_csg_scene(300, 200)
    gs, cs = {}, {}
    img = world.render(1, stats=gs)
    cpu = oracle.rtc_render(world.desc, world.camera, aa=1, stats=cs)
    for k in COUNTERS:
        assert gs[k] == cs[k], (k, gs[k], cs[k])
    assert np.abs(img - cpu).max() <= TOL and np.abs(img - cpu).max() <= 1e-12
    assert rl.canvas_ppm(img).encode() == golden("test_csg_scene.ppm.gz")


def test_rtc_mirror_scene_on_gpu(rl, oracle, golden):
    world = rl.RtcWorld.test_mirror_scene(300, 200)
    gs, cs = {}, {}
    img = world.render(1, stats=gs)
    cpu = oracle.rtc_render(world.desc, world.camera, aa=1, stats=cs)
    for k in COUNTERS:
        assert gs[k] == cs[k], (k, gs[k], cs[k])
    # weighted-sum form of the reflection / refraction tree + device pow(): ulp-level colour differences only
    assert np.abs(img - cpu).max() <= TOL and np.abs(img - cpu).max() <= 1e-11
    got, want = _ppm_array(rl.canvas_ppm(img)), _ppm_array(golden("test_mirror_scene.ppm.gz").decode())
    ys, xs = np.nonzero((got != want).any(axis=2))
    assert len(ys) <= 1 and all((x, y) == (23, 95) for x, y in zip(xs, ys))  # see tests/test_oracle_golden.py


def test_rtc_cylinders_cones_patterns_vs_oracle(rl, oracle):
    api = rl.api

    def mat(color=(1, 1, 1), pattern=0, **kw):
        m = np.zeros(1, dtype=api.RTC_MATERIAL)
        m["color"], m["ambient"], m["diffuse"], m["specular"], m["shininess"], m["refractive_index"], m["pattern"] = color, 0.1, 0.9, 0.9, 200.0, 1.0, pattern
        for k, v in kw.items():
            m[k] = v
        return m[0]

    def xf(matrix, kind, index):
        return api.rtc_transformed(matrix, kind, index)

    T = lambda x, y, z: np.array([[1, 0, 0, x], [0, 1, 0, y], [0, 0, 1, z], [0, 0, 0, 1]], dtype=float)
    S = lambda x, y, z: np.diag([x, y, z, 1.0])
    pats = np.zeros(3, dtype=api.RTC_PATTERN)
    pats[0]["kind"], pats[0]["a"], pats[0]["b"], pats[0]["inverse"] = api.PAT_RING, (1, 0.3, 0.3), (0.2, 0.2, 1), np.linalg.inv(S(0.5, 0.5, 0.5)).reshape(16)
    pats[1]["kind"], pats[1]["a"], pats[1]["b"], pats[1]["inverse"] = api.PAT_GRADIENT, (1, 1, 0), (0, 1, 1), np.eye(4).reshape(16)
    pats[2]["kind"], pats[2]["a"], pats[2]["b"], pats[2]["inverse"] = api.PAT_STRIPE, (1, 1, 1), (0.1, 0.1, 0.1), np.linalg.inv(S(0.3, 1, 1)).reshape(16)
    mats = np.array([mat(pattern=1, reflectivity=0.1), mat(pattern=2), mat(pattern=3), mat((0.2, 0.9, 0.3), reflectivity=0.4),
                     mat((0.9, 0.9, 1.0), diffuse=0.1, transparency=0.9, reflectivity=0.9, refractive_index=1.5)], dtype=api.RTC_MATERIAL)
    shapes = np.zeros(5, dtype=api.RTC_SHAPE)
    shapes[0]["kind"], shapes[0]["material"] = api.O_PLANE, 0
    shapes[1]["kind"], shapes[1]["material"], shapes[1]["has_minimum"], shapes[1]["has_maximum"], shapes[1]["minimum"], shapes[1]["maximum"], shapes[1]["closed"] = api.O_CYLINDER, 1, 1, 1, 0.0, 2.0, 1
    shapes[2]["kind"], shapes[2]["material"], shapes[2]["has_minimum"], shapes[2]["has_maximum"], shapes[2]["minimum"], shapes[2]["maximum"], shapes[2]["closed"] = api.O_CONE, 2, 1, 1, -1.0, 0.0, 1
    shapes[3]["kind"], shapes[3]["material"] = api.O_CUBE, 3
    shapes[4]["kind"], shapes[4]["material"] = api.O_SPHERE, 4
    tr = np.array([xf(T(-2.5, 0, 1), api.O_CYLINDER, 1), xf(T(0, 1.0, 0) @ S(1, 1.5, 1), api.O_CONE, 2), xf(T(2.6, 0.5, 1.5) @ S(0.5, 0.5, 0.5), api.O_CUBE, 3),
                   xf(T(0.8, 0.8, -1.8) @ S(0.8, 0.8, 0.8), api.O_SPHERE, 4)], dtype=api.RTC_TRANSFORMED)
    objs = np.zeros(5, dtype=api.HREF)
    objs["kind"] = [api.O_PLANE, api.O_TRANSFORMED, api.O_TRANSFORMED, api.O_TRANSFORMED, api.O_TRANSFORMED]
    objs["index"] = [0, 0, 1, 2, 3]
    lights = np.zeros(2, dtype=api.RTC_LIGHT)
    lights["position"], lights["intensity"] = [(-5, 8, -8), (6, 5, -4)], [(0.7, 0.7, 0.7), (0.4, 0.4, 0.3)]
    cam = api.rtc_camera(160, 100, 1.0, (0, 2.5, -8), (0, 0.8, 0), (0, 1, 0))
    world = rl.RtcWorld.from_arrays(np.zeros(0, dtype=api.RTC_TRIANGLE), mats, objs, lights, transformeds=tr, shapes=shapes, patterns=pats, camera=cam)
    gs, cs = {}, {}
    img = world.render(2, stats=gs)
    cpu = oracle.rtc_render(world.desc, world.camera, aa=2, stats=cs)
    for k in COUNTERS:
        assert gs[k] == cs[k], (k, gs[k], cs[k])
    assert np.abs(img - cpu).max() <= TOL and np.abs(img - cpu).max() <= 1e-11
    assert img.max() > 0.2


def test_rtc_reflection_depth_cap_is_loud_and_depth_7_matches_oracle(rl, oracle):
    """A material that is reflective AND transparent spawns two rays per hit, so the kernel's depth-first walk needs
    max_reflection_depth + 1 pending slots (RTC_MAX_PENDING = 8).  Depth 7 — the deepest world the library accepts — must match the
    oracle's recursion (world.rs:128-159) counter for counter; depth 8 must be refused with RL_E_UNSUPPORTED, never truncated."""
    api = rl.api
    m = np.zeros(3, dtype=api.RTC_MATERIAL)
    m["ambient"], m["diffuse"], m["specular"], m["shininess"], m["refractive_index"] = 0.1, 0.3, 0.9, 200.0, 1.0
    m[0]["color"], m[0]["reflectivity"], m[0]["transparency"], m[0]["refractive_index"] = (0.9, 0.9, 1.0), 0.6, 0.8, 1.5  # glass-mirror
    m[1]["color"], m[1]["reflectivity"], m[1]["transparency"], m[1]["refractive_index"] = (1.0, 0.8, 0.8), 0.5, 0.7, 1.3
    m[2]["color"], m[2]["reflectivity"] = (0.3, 0.6, 0.3), 0.5
    shapes = np.zeros(4, dtype=api.RTC_SHAPE)
    shapes["kind"], shapes["material"] = [api.O_PLANE, api.O_SPHERE, api.O_SPHERE, api.O_CUBE], [2, 0, 1, 0]
    T = lambda x, y, z: np.array([[1, 0, 0, x], [0, 1, 0, y], [0, 0, 1, z], [0, 0, 0, 1]], dtype=float)
    S = lambda x, y, z: np.diag([x, y, z, 1.0])
    tr = np.array([api.rtc_transformed(T(-1.1, 1, 0), api.O_SPHERE, 1), api.rtc_transformed(T(-1.1, 1, 0) @ S(0.5, 0.5, 0.5), api.O_SPHERE, 2),
                   api.rtc_transformed(T(1.3, 0.8, 0.5) @ S(0.8, 0.8, 0.8), api.O_CUBE, 3)], dtype=api.RTC_TRANSFORMED)
    objs = np.zeros(4, dtype=api.HREF)
    objs["kind"], objs["index"] = [api.O_PLANE, api.O_TRANSFORMED, api.O_TRANSFORMED, api.O_TRANSFORMED], [0, 0, 1, 2]
    lights = np.zeros(2, dtype=api.RTC_LIGHT)
    lights["position"], lights["intensity"] = [(-5, 8, -8), (6, 5, -4)], [(0.7, 0.7, 0.7), (0.4, 0.4, 0.3)]
    cam = api.rtc_camera(96, 64, 1.0, (0, 2.0, -6), (0, 0.8, 0), (0, 1, 0))

    def build(depth):
        return rl.RtcWorld.from_arrays(np.zeros(0, dtype=api.RTC_TRIANGLE), m, objs, lights, transformeds=tr, shapes=shapes, camera=cam,
                                       max_reflection_depth=depth, void_color=(0.1, 0.1, 0.2))

    world = build(7)
    gs, cs = {}, {}
    img = world.render(1, stats=gs)
    cpu = oracle.rtc_render(world.desc, world.camera, aa=1, stats=cs)
    for k in COUNTERS:
        assert gs[k] == cs[k], (k, gs[k], cs[k])
    assert gs["rays"] > 40 * 96 * 64  # the tree really branches: two lights x two sub-rays per hit
    assert np.abs(img - cpu).max() <= 1e-10
    with pytest.raises(api.RLError) as e:
        build(8).render(1)
    assert "max_reflection_depth" in str(e.value), str(e.value)


def test_rtc_reflective_mesh_through_the_full_kernel_walks_the_guard_tree(rl, oracle, golden):
    """The teapot of tests/ray_tracer.rs:242-275 with a reflective, slightly transparent material: a triangle mesh through
    World::color_at's full recursion.  The full kernel walks the reject-only box tree over each triangle range with an UNBOUNDED line
    test (it needs every intersection, of either sign of t); counters and colours must equal the oracle's every-triangle loop."""
    import ctypes as C
    api = rl.api
    world = rl.RtcWorld.test_obj_scene(golden("teapot-low.obj"), 120, 80)
    desc = api.RtcSceneDesc.from_address(world.desc)
    mats = np.frombuffer((C.c_char * (desc.n_materials * api.RTC_MATERIAL.itemsize)).from_address(desc.materials), dtype=api.RTC_MATERIAL)
    mats["reflectivity"], mats["transparency"], mats["refractive_index"] = 0.35, 0.2, 1.3
    gs, cs = {}, {}
    img = world.render(1, stats=gs)
    cpu = oracle.rtc_render(world.desc, world.camera, aa=1, stats=cs)
    for k in COUNTERS:
        assert gs[k] == cs[k], (k, gs[k], cs[k])
    assert gs["rays"] > 3 * 120 * 80 and gs["planar_tests"] > 200 * gs["rays"] // 2  # secondary rays, every triangle counted
    assert np.abs(img - cpu).max() <= 1e-10
