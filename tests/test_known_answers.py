"""CPU tier: the reference's own unit known-answers for this path, restated as VALUES
and checked against the oracle's probes.  Together with the two golden images these pin the oracle.

  ChaCha8 / rand            SURVEY.md A.1 self-check values (rand_chacha 0.3.1, rand 0.8.5)
  Sphere::hit               RTIOW src/hittable/sphere.rs:118-179
  slice closest hit         RTIOW src/hittable/mod.rs:226-250
  AABB::hit                 RTIOW src/aabb.rs:182-205
  get_sphere_uv             RTIOW src/hittable/sphere.rs:181-207
  Triangle::intersect       RTC  src/scene/object/triangle.rs:170-251
  lighting                  RTC  src/scene/material.rs:136-274
"""
import math

import numpy as np
import pytest

INF = float("inf")


# ----------------------------------------------------------------------------- ChaCha8 (A.1)
def test_chacha_key_expansion(oracle):
    assert [f"{w:08x}" for w in oracle.chacha_key(0)] == "f973f2ec 45cdb581 7346f087 ad6cad06 e3a3d0d0 67e71733 72ea9bf2 fe7d8ad7".split()
    assert [f"{w:08x}" for w in oracle.chacha_key(1)] == "721dd8ea 4e10265d f83b9c89 2e78ce42 da03d3ba c2d29799 ac560212 1bfb6673".split()


def test_chacha_blocks(oracle):
    assert [f"{w:08x}" for w in oracle.chacha_block(0, 0, 0)[:4]] == "a79a3b6c b585f767 bad8c037 7746a55f".split()
    assert [f"{w:08x}" for w in oracle.chacha_block(0, 0, 5)[:4]] == "eb2cc2ba 2d902c67 b22c5c3b 2f1995de".split()
    assert [f"{w:08x}" for w in oracle.chacha_block(0, 1, 5)[:2]] == "3aaac993 680fb15e".split()


def test_chacha_draws_and_set_stream_keeps_position(oracle):
    out, pos = oracle.chacha_script(0, [("f64",), ("f64",), ("uniform",)])
    assert out[0] == 0.7090754154265618 and out[1] == 0.46592172228961015 and out[2] == 0.3982864853494634
    assert pos == 6
    out, pos = oracle.chacha_script(0, [("f64",), ("set_stream", 7), ("f64",)])
    assert out[2] == 0.06264332974516373 and pos == 4  # pos stayed 2 across set_stream


# ----------------------------------------------------------------------------- RTIOW geometry
def _sphere_world(rl, spheres, use_bvh=False):
    api = rl.api
    tex = np.zeros(1, dtype=api.TEXTURE)
    mats = np.zeros(1, dtype=api.MATERIAL)  # Flat
    sph = np.zeros(len(spheres), dtype=api.SPHERE)
    for i, (c, r) in enumerate(spheres):
        sph[i]["center0"], sph[i]["radius"] = c, r
    return rl.World.from_spheres(sph, mats, tex, use_bvh)


def test_sphere_hit_known_answers(rl, oracle):
    w = _sphere_world(rl, [((0, 0, 0), 1.0)])
    assert oracle.rtiow_hit(w.desc, (0, 2, 5), (0, 0, -1)) is None  # misses
    h = oracle.rtiow_hit(w.desc, (0, 1, 5), (0, 0, -1))  # tangent
    assert h["t"] == 5.0 and h["front"] and np.allclose(h["normal"], (0, 1, 0))
    h = oracle.rtiow_hit(w.desc, (0, 0, 5), (0, 0, -1))  # through
    assert h["t"] == 4.0 and h["front"] and np.allclose(h["normal"], (0, 0, 1))
    h = oracle.rtiow_hit(w.desc, (0, 0, 0), (0, 0, -1))  # from inside
    assert h["t"] == 1.0 and not h["front"] and np.allclose(h["normal"], (0, 0, 1))
    assert oracle.rtiow_hit(w.desc, (0, 0, 5), (0, 0, -1), tmin=0.0, tmax=1.0) is None
    assert oracle.rtiow_hit(w.desc, (0, 0, 5), (0, 0, -1), tmin=0.0, tmax=4.0)["t"] == 4.0  # interval is closed


def test_slice_returns_closest_of_three(rl, oracle):
    w = _sphere_world(rl, [((0, 0, -10), 1.0), ((0, 0, 0), 1.0), ((0, 0, -5), 1.0)])
    assert oracle.rtiow_hit(w.desc, (0, 0, 5), (0, 0, -1))["t"] == 4.0
    wb = _sphere_world(rl, [((0, 0, -10), 1.0), ((0, 0, 0), 1.0), ((0, 0, -5), 1.0)], use_bvh=True)
    assert oracle.rtiow_hit(wb.desc, (0, 0, 5), (0, 0, -1))["t"] == 4.0


def test_equal_t_later_hittable_wins(rl, oracle):
    # two coincident spheres: the fold replaces on t <= closest, so the LATER one is reported (hittable/mod.rs:90-105)
    api = rl.api
    tex = np.zeros(1, dtype=api.TEXTURE)
    mats = np.zeros(2, dtype=api.MATERIAL)
    sph = np.zeros(2, dtype=api.SPHERE)
    sph["center0"], sph["radius"], sph["material"] = (0, 0, 0), 1.0, (0, 1)
    w = rl.World.from_spheres(sph, mats, tex, False)
    assert oracle.rtiow_hit(w.desc, (0, 0, 5), (0, 0, -1))["mat"] == 1


@pytest.mark.parametrize("o,d,tmin,tmax,expected", [
    ((0, 3, 3), (1, 0, 0), -INF, INF, True), ((3, 5, 3), (0, -1, 0), -INF, INF, True), ((3, 3, 1.5), (0, 0, 1), -INF, INF, True),
    ((5, 5, 5), (0, 0, 1), -INF, INF, False), ((0, 3, 3), (1, 0, 0), 2.0, 3.0, True), ((0, 3, 3), (1, 0, 0), 1.0, 1.9, False)])
def test_aabb_hit_known_answers(oracle, o, d, tmin, tmax, expected):
    assert oracle.aabb_hit((2, 4, 2, 4, 2, 4), o, d, tmin, tmax) is expected


@pytest.mark.parametrize("p,uv", [((1, 0, 0), (0.5, 0.5)), ((0, 1, 0), (0.5, 1.0)), ((0, 0, 1), (0.25, 0.5)),
                                  ((-1, 0, 0), (0.0, 0.5)), ((0, -1, 0), (0.5, 0.0)), ((0, 0, -1), (0.75, 0.5))])
def test_sphere_uv_known_answers(oracle, p, uv):  # sphere.rs:198-205 (epsilon 0.01 there)
    u, v = oracle.sphere_uv(p)
    assert abs(u - uv[0]) <= 0.01 and abs(v - uv[1]) <= 0.01


def test_sphere_hit_record_carries_uv(rl, oracle):  # sphere.rs:70: uv of the outward normal at the hit point
    world = rl.World.build(lambda b: b.sphere((0, 0, 0), 2.0, b.lambertian(b.solid((1, 1, 1)))))
    h = oracle.rtiow_hit(world.desc, (0, 0, 12), (0, 0, -1), 0.0, 1e-10, INF)   # hits (0, 0, 2): the +z pole of get_sphere_uv
    assert h is not None and abs(h["t"] - 10.0) < 1e-12 and abs(h["u"] - 0.25) < 1e-12 and abs(h["v"] - 0.5) < 1e-12


def test_perlin_noise_properties(rl, oracle):
    """perlin.rs has no unit tests; these are properties of its algorithm: the Hermite-blended gradient noise vanishes on
    the integer lattice, is periodic with the 256-cell permutation tables, stays inside [-1, 1] ("Outputs a noise value ...
    in the range [-1, 1]", perlin.rs:38) and turb() is the |sum of 7 halved octaves|."""
    rng = np.random.default_rng(5)
    pn = np.zeros(1, dtype=rl.api.PERLIN)
    v = rng.normal(size=(256, 3))
    pn["randvec"][0] = v / np.linalg.norm(v, axis=1, keepdims=True)
    for k in ("perm_x", "perm_y", "perm_z"):
        pn[k][0] = rng.permutation(256)
    for p in ((0, 0, 0), (3, -7, 11), (255, 256, -256)):
        assert oracle.perlin_noise(pn, p) == 0.0
    pts = rng.uniform(-40, 40, size=(200, 3))
    vals = np.array([oracle.perlin_noise(pn, p) for p in pts])
    assert np.abs(vals).max() <= 1.0 and np.abs(vals).max() > 0.2
    for p in pts[:20]:
        assert abs(oracle.perlin_noise(pn, p + np.array([256.0, -512.0, 768.0])) - oracle.perlin_noise(pn, p)) < 1e-9
        acc, w, q = 0.0, 1.0, np.array(p)
        for _ in range(7):
            acc += w * oracle.perlin_noise(pn, q)
            w *= 0.5
            q = q * 2.0
        assert oracle.perlin_turb(pn, p, 7) == abs(acc)


# ----------------------------------------------------------------------------- RTC
def _tri_world(rl, smooth=False):
    api = rl.api
    t = np.zeros(1, dtype=api.RTC_TRIANGLE)
    p1, p2, p3 = np.array([0, 1, 0.0]), np.array([-1, 0, 0.0]), np.array([1, 0, 0.0])
    t["p1"], t["e1"], t["e2"] = p1, p2 - p1, p3 - p1
    if smooth:
        t["smooth"], t["n1"], t["n2"], t["n3"] = 1, (0, 1, 0), (-1, 0, 0), (1, 0, 0)
    else:
        t["n1"] = (0, 0, -1)
    m = np.zeros(1, dtype=api.RTC_MATERIAL)
    m["color"], m["ambient"], m["diffuse"], m["specular"], m["shininess"], m["refractive_index"] = (1, 1, 1), 0.1, 0.9, 0.9, 200.0, 1.0
    objs = np.zeros(1, dtype=api.HREF)
    objs["kind"], objs["index"] = api.O_TRIANGLE, 0
    return rl.RtcWorld.from_arrays(t, m, objs, np.zeros(0, dtype=api.RTC_LIGHT)), m


def test_rtc_triangle_intersect_known_answers(rl, oracle):
    w, _ = _tri_world(rl)
    assert len(oracle.rtc_intersect(w.desc, (0, -1, -2), (0, 1, 0))[0]) == 0  # parallel
    assert len(oracle.rtc_intersect(w.desc, (1, 1, -2), (0, 0, 1))[0]) == 0  # misses p1-p3
    assert len(oracle.rtc_intersect(w.desc, (-1, 1, -2), (0, 0, 1))[0]) == 0  # misses p1-p2
    assert len(oracle.rtc_intersect(w.desc, (0, -1, -2), (0, 0, 1))[0]) == 0  # misses p2-p3
    ts, objs, ns = oracle.rtc_intersect(w.desc, (0, 0.5, -2), (0, 0, 1))
    assert list(ts) == [2.0] and tuple(ns[0]) == (0.0, 0.0, -1.0)


def test_rtc_smooth_triangle_interpolates_normals(rl, oracle):
    w, _ = _tri_world(rl, smooth=True)
    ts, objs, ns = oracle.rtc_intersect(w.desc, (-0.2, 0.3, -2), (0, 0, 1))
    assert np.allclose(ns[0], (-0.55470, 0.83205, 0.0), atol=1e-5)


def test_rtc_lighting_six_cases(rl, oracle):
    _, m = _tri_world(rl)
    t = math.sqrt(2.0) / 2.0
    white, origin, n = (1, 1, 1), (0, 0, 0), (0, 0, -1)
    L = lambda lp, eye, att=1.0: oracle.rtc_lighting(m[0], origin, lp, white, eye, n, att)
    assert tuple(L((0, 0, -10), (0, 0, -1))) == (1.9, 1.9, 1.9)
    assert tuple(L((0, 0, -10), (0, t, -t))) == (1.0, 1.0, 1.0)
    assert np.allclose(L((0, 10, -10), (0, 0, -1)), 0.7364, atol=1e-5)
    assert np.allclose(L((0, 10, -10), (0, -t, -t)), 1.6364, atol=1e-5)
    assert tuple(L((0, 0, 10), (0, 0, -1))) == (0.1, 0.1, 0.1)
    assert tuple(L((0, 0, -10), (0, 0, -1), 0.0)) == (0.1, 0.1, 0.1)


def test_oracle_bvh_new_known_structure(oracle):
    """Bvh::new (bvh.rs:22-60) + find_longest_axis (bvh.rs:63-77) on five unit boxes spread along x, by hand:
    root box = union, longest axis x -> sorted by x.min -> [0 1 | 2 3 4]; left = leaf(0, 1); right: 3 boxes -> [2 | 3 4]."""
    centres = np.array([[8.0, 0, 0], [0.0, 0, 0], [4.0, 0.5, 0], [2.0, 0, 0.25], [6.0, 0, 0]])
    boxes = np.empty((5, 6))
    boxes[:, 0::2], boxes[:, 1::2] = centres - 0.5, centres + 0.5
    prims = np.zeros(5, dtype=[("kind", "<u4"), ("index", "<u4")])
    prims["kind"], prims["index"] = 1, np.arange(5) + 10
    nodes = oracle.bvh_build(boxes, prims, node_base=100)
    # creation order: root, left(leaf 1,3), right, right.left(leaf 2), right.right(leaf 4,0)
    assert len(nodes) == 5
    root, left, right, rl_, rr = nodes
    assert list(root["bbox"]) == [-0.5, 8.5, -0.5, 1.0, -0.5, 0.75]
    assert [tuple(c) for c in root["child"]] == [(5, 101), (5, 102)]
    assert left["n_children"] == 2 and [tuple(c) for c in left["child"]] == [(1, 11), (1, 13)]
    assert list(left["bbox"]) == [-0.5, 2.5, -0.5, 0.5, -0.5, 0.75]
    assert [tuple(c) for c in right["child"]] == [(5, 103), (5, 104)] and list(right["bbox"])[:2] == [3.5, 8.5]
    assert rl_["n_children"] == 1 and tuple(rl_["child"][0]) == (1, 12)
    assert rr["n_children"] == 2 and [tuple(c) for c in rr["child"]] == [(1, 14), (1, 10)]
    # y longest only when strictly larger than x AND z (bvh.rs:64-76): equal sizes fall through to z
    eq = np.array([[0.0, 1, 0, 1, 0, 1], [0.0, 1, 0, 1, 5, 6], [0.0, 1, 0, 1, 2, 3]])
    eq = np.vstack([eq, [0.0, 6, 0, 6, 0, 1]])  # union: x 6, y 6, z 6 -> not x (6 > 6 false), not y -> z
    p4 = np.zeros(4, dtype=prims.dtype)
    p4["kind"], p4["index"] = 1, np.arange(4)
    n4 = oracle.bvh_build(eq, p4)
    assert [tuple(c) for c in n4[1]["child"]] == [(1, 0), (1, 3)]  # sorted by z.min, stable: boxes 0 and 3 (z.min 0) first
    assert [tuple(c) for c in n4[2]["child"]] == [(1, 2), (1, 1)]
