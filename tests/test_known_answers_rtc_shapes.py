"""CPU tier: the reference's unit known-answers for the RTC analytic shapes, CSG and the reflection / refraction paths of
World::color_at, restated as VALUES and checked against the oracle's probes.

  Sphere::intersect / normal_at      RTC src/scene/object/sphere.rs:96-215
  Plane                              RTC src/scene/object/plane.rs:57-115
  Cube                               RTC src/scene/object/cube.rs (cube_intersect_tests!, cube_normal_tests!)
  Cylinder                           RTC src/scene/object/cylinder.rs (intersect / normal / truncate / closed tables)
  Cone                               RTC src/scene/object/cone.rs (intersect / endcap / normal tables)
  Csg                                RTC src/scene/object/csg.rs:172-211
  World::color_at, shade_hit         RTC src/scene/world.rs:240-350,486-560,665-750

The reference compares intersection distances with assert_eq! (exact f64 equality): so do these tests, which pins the
oracle's order of operations, not just its formulas.  Colours are compared with the reference's own tolerance (1e-5).
"""
import math

import numpy as np
import pytest

SQ2 = math.sqrt(2.0)


def _norm(v):  # Vec3d::norm (math/vector.rs:32-43): component / sqrt(x*x + y*y + z*z)
    x, y, z = (float(c) for c in v)
    m = math.sqrt(x * x + y * y + z * z)
    return (x / m, y / m, z / m)


def _material(api, **kw):
    m = np.zeros(1, dtype=api.RTC_MATERIAL)
    m["color"], m["ambient"], m["diffuse"], m["specular"], m["shininess"], m["refractive_index"] = (1, 1, 1), 0.1, 0.9, 0.9, 200.0, 1.0
    for k, v in kw.items():
        m[k] = v
    return m[0]


def _shape_world(rl, kind, minimum=None, maximum=None, closed=False):
    api = rl.api
    sh = np.zeros(1, dtype=api.RTC_SHAPE)
    sh["kind"], sh["material"], sh["closed"] = kind, 0, 1 if closed else 0
    if minimum is not None:
        sh["has_minimum"], sh["minimum"] = 1, minimum
    if maximum is not None:
        sh["has_maximum"], sh["maximum"] = 1, maximum
    objs = np.zeros(1, dtype=api.HREF)
    objs["kind"], objs["index"] = kind, 0
    mats = np.array([_material(api)], dtype=api.RTC_MATERIAL)
    return rl.RtcWorld.from_arrays(np.zeros(0, dtype=api.RTC_TRIANGLE), mats, objs, np.zeros(0, dtype=api.RTC_LIGHT), shapes=sh)


def _ts(oracle, world, o, d):
    return [float(t) for t in oracle.rtc_intersect(world.desc, o, d)[0]]


# ----------------------------------------------------------------------------- sphere / plane
def test_sphere_intersections(rl, oracle):
    w = _shape_world(rl, rl.api.O_SPHERE)
    assert _ts(oracle, w, (0, 0, -5), (0, 0, 1)) == [4.0, 6.0]
    assert _ts(oracle, w, (0, 1, -5), (0, 0, 1)) == [5.0, 5.0]  # tangent: two equal intersections
    assert _ts(oracle, w, (0, 2, -5), (0, 0, 1)) == []
    assert _ts(oracle, w, (0, 0, 0), (0, 0, 1)) == [-1.0, 1.0]  # ray starts inside
    assert _ts(oracle, w, (0, 0, 5), (0, 0, 1)) == [-6.0, -4.0]  # sphere behind the ray


def test_sphere_normals(rl, oracle):
    w = _shape_world(rl, rl.api.O_SPHERE)
    for o, d, n in (((5, 0, 0), (-1, 0, 0), (1, 0, 0)), ((0, 5, 0), (0, -1, 0), (0, 1, 0)), ((0, 0, 5), (0, 0, -1), (0, 0, 1))):
        _, _, normals = oracle.rtc_intersect(w.desc, o, d)
        assert np.allclose(normals[0], n, atol=1e-15)
    t = math.sqrt(3.0) / 3.0
    _, _, normals = oracle.rtc_intersect(w.desc, (5 * t, 5 * t, 5 * t), (-t, -t, -t))
    assert np.allclose(normals[0], (t, t, t), atol=1e-12) and abs(np.linalg.norm(normals[0]) - 1.0) < 1e-15


def test_plane(rl, oracle):
    w = _shape_world(rl, rl.api.O_PLANE)
    assert _ts(oracle, w, (0, 10, 0), (0, 0, 1)) == []  # parallel
    assert _ts(oracle, w, (0, 0, 0), (0, 0, 1)) == []   # coplanar
    assert _ts(oracle, w, (0, 1, 0), (0, -1, 0)) == [1.0]
    assert _ts(oracle, w, (0, -1, 0), (0, 1, 0)) == [1.0]
    _, _, normals = oracle.rtc_intersect(w.desc, (10, 1, -10), (0, -1, 0))
    assert tuple(normals[0]) == (0.0, 1.0, 0.0)


# ----------------------------------------------------------------------------- cube
@pytest.mark.parametrize("o,d,ts", [
    ((5, 0.5, 0), (-1, 0, 0), [4.0, 6.0]), ((-5, 0.5, 0), (1, 0, 0), [4.0, 6.0]), ((0.5, 5, 0), (0, -1, 0), [4.0, 6.0]),
    ((0.5, -5, 0), (0, 1, 0), [4.0, 6.0]), ((0.5, 0, 5), (0, 0, -1), [4.0, 6.0]), ((0.5, 0, -5), (0, 0, 1), [4.0, 6.0]),
    ((0, 0.5, 0), (0, 0, 1), [-1.0, 1.0]),
    ((-2, 0, 0), (0.2673, 0.5345, 0.8018), []), ((0, -2, 0), (0.8018, 0.2673, 0.5345), []), ((0, 0, -2), (0.5345, 0.8018, 0.2673), []),
    ((2, 0, 2), (0, 0, -1), []), ((0, 2, 2), (0, -1, 0), []), ((2, 2, 0), (-1, 0, 0), [])])
def test_cube_intersections(rl, oracle, o, d, ts):
    assert _ts(oracle, _shape_world(rl, rl.api.O_CUBE), o, d) == ts


@pytest.mark.parametrize("o,d,n", [((5, 0.5, -0.8), (-1, 0, 0), (1, 0, 0)), ((-5, -0.2, 0.9), (1, 0, 0), (-1, 0, 0)), ((-0.4, 5, -0.1), (0, -1, 0), (0, 1, 0)),
                                   ((0.3, -5, -0.7), (0, 1, 0), (0, -1, 0)), ((-0.6, 0.3, 5), (0, 0, -1), (0, 0, 1)), ((0.4, 0.4, -5), (0, 0, 1), (0, 0, -1))])
def test_cube_normals(rl, oracle, o, d, n):  # cube_normal_1..6: the first hit of an axis ray lands on the tabulated point
    _, _, normals = oracle.rtc_intersect(_shape_world(rl, rl.api.O_CUBE).desc, o, d)
    assert tuple(normals[0]) == tuple(float(c) for c in n)


# ----------------------------------------------------------------------------- cylinder
@pytest.mark.parametrize("o,d,ts", [
    ((1, 0, 0), (0, 1, 0), []), ((0, 0, 0), (0, 1, 0), []), ((0, 0, -5), (1, 1, 1), []),
    ((1, 0, -5), (0, 0, 1), [5.0, 5.0]), ((0, 0, -5), (0, 0, 1), [4.0, 6.0]), ((0.5, 0, -5), (0.1, 1, 1), [6.80798191702732, 7.088723439378861])])
def test_cylinder_intersections(rl, oracle, o, d, ts):
    assert _ts(oracle, _shape_world(rl, rl.api.O_CYLINDER), o, _norm(d)) == ts


@pytest.mark.parametrize("o,d,count", [((0, 1.5, 0), (0.1, 1, 0), 0), ((0, 3, -5), (0, 0, 1), 0), ((0, 0, -5), (0, 0, 1), 0),
                                       ((0, 2, -5), (0, 0, 1), 0), ((0, 1, -5), (0, 0, 1), 0), ((0, 1.5, -2), (0, 0, 1), 2)])
def test_truncated_cylinder(rl, oracle, o, d, count):
    assert len(_ts(oracle, _shape_world(rl, rl.api.O_CYLINDER, 1.0, 2.0), o, _norm(d))) == count


@pytest.mark.parametrize("o,d,count", [((0, 3, 0), (0, -1, 0), 2), ((0, 3, -2), (0, -1, 2), 2), ((0, 4, -2), (0, -1, 1), 2),
                                       ((0, 0, -2), (0, 1, 2), 2), ((0, -1, -2), (0, 1, 1), 2)])
def test_capped_cylinder(rl, oracle, o, d, count):
    assert len(_ts(oracle, _shape_world(rl, rl.api.O_CYLINDER, 1.0, 2.0, closed=True), o, _norm(d))) == count


def test_cylinder_normals(rl, oracle):
    w = _shape_world(rl, rl.api.O_CYLINDER)
    for o, d, n in (((5, 0, 0), (-1, 0, 0), (1, 0, 0)), ((0, 5, -5), (0, 0, 1), (0, 0, -1)), ((0, -2, 5), (0, 0, -1), (0, 0, 1)), ((-5, 1, 0), (1, 0, 0), (-1, 0, 0))):
        _, _, normals = oracle.rtc_intersect(w.desc, o, d)
        assert tuple(normals[0]) == tuple(float(c) for c in n)
    capped = _shape_world(rl, rl.api.O_CYLINDER, 1.0, 2.0, closed=True)  # cylinder_cap_normal_tests!: caps point along -y / +y
    _, _, normals = oracle.rtc_intersect(capped.desc, (0.5, 5, 0), (0, -1, 0))
    assert tuple(normals[0]) == (0.0, 1.0, 0.0) and tuple(normals[1]) == (0.0, -1.0, 0.0)


# ----------------------------------------------------------------------------- cone
@pytest.mark.parametrize("o,d,ts", [((0, 1e-6, -5), (0, 0, 1), [4.999999000844085, 5.000000999155915]),
                                    ((0, 0, -5), (1, 1, 1), [8.660254037844386, 8.660254037844386]),
                                    ((1, 1, -5), (-0.5, -1, 1), [4.550055679356349, 49.449944320643645]),
                                    ((0, 0, -1), (0, 1, 1), [0.3535533905932738])])  # last: parallel to one half -> one intersection
def test_cone_intersections(rl, oracle, o, d, ts):
    assert _ts(oracle, _shape_world(rl, rl.api.O_CONE), o, _norm(d)) == ts


@pytest.mark.parametrize("o,d,count", [((0, 0, -5), (0, 1, 0), 0), ((0, 0, -0.25), (0, 1, 1), 2), ((0, 0, -0.25), (0, 1, 0), 4)])
def test_cone_end_caps(rl, oracle, o, d, count):
    assert len(_ts(oracle, _shape_world(rl, rl.api.O_CONE, -0.5, 0.5, closed=True), o, _norm(d))) == count


def test_cone_normal(rl, oracle):  # cone_normal_tests!: n(-1,-1,0) = norm(-1, 1, 0); reached by a ray along +x at y = -1
    _, _, normals = oracle.rtc_intersect(_shape_world(rl, rl.api.O_CONE).desc, (-5, -1, 0), (1, 0, 0))
    assert np.allclose(normals[0], _norm((-1, 1, 0)), atol=1e-15)


# ----------------------------------------------------------------------------- CSG
def _csg_world(rl, op, right_kind, right_translate_z=None):
    api = rl.api
    sh = np.zeros(2, dtype=api.RTC_SHAPE)
    sh["kind"], sh["material"] = [api.O_SPHERE, right_kind], [0, 0]
    tr = np.zeros(0, dtype=api.RTC_TRANSFORMED)
    right = (right_kind, 1)
    if right_translate_z is not None:
        T = np.eye(4)
        T[2, 3] = right_translate_z
        tr = np.array([api.rtc_transformed(T, right_kind, 1)], dtype=api.RTC_TRANSFORMED)
        right = (api.O_TRANSFORMED, 0)
    csg = np.zeros(1, dtype=api.RTC_CSG)
    csg["operation"] = op
    csg["left"]["kind"], csg["left"]["index"] = api.O_SPHERE, 0
    csg["right"]["kind"], csg["right"]["index"] = right
    objs = np.zeros(1, dtype=api.HREF)
    objs["kind"], objs["index"] = api.O_CSG, 0
    mats = np.array([_material(api)], dtype=api.RTC_MATERIAL)
    return rl.RtcWorld.from_arrays(np.zeros(0, dtype=api.RTC_TRIANGLE), mats, objs, np.zeros(0, dtype=api.RTC_LIGHT), shapes=sh, csgs=csg, transformeds=tr)


def test_csg_known_answers(rl, oracle):
    api = rl.api
    assert _ts(oracle, _csg_world(rl, api.CSG_UNION, api.O_CUBE), (0, 2, -5), (0, 0, 1)) == []  # a_ray_misses_a_csg_object
    # a_ray_hits_a_csg_object: sphere U sphere translated by z = 0.5 -> the left's first and the right's second intersection
    assert _ts(oracle, _csg_world(rl, api.CSG_UNION, api.O_SPHERE, 0.5), (0, 0, -5), (0, 0, 1)) == [4.0, 6.5]
    # the rule table (csg.rs:130-170) through geometry: intersection keeps [4.5, 6], difference keeps [4, 4.5]
    assert _ts(oracle, _csg_world(rl, api.CSG_INTERSECTION, api.O_SPHERE, 0.5), (0, 0, -5), (0, 0, 1)) == [4.5, 6.0]
    assert _ts(oracle, _csg_world(rl, api.CSG_DIFFERENCE, api.O_SPHERE, 0.5), (0, 0, -5), (0, 0, 1)) == [4.0, 4.5]


# ----------------------------------------------------------------------------- World::basic() and color_at
def _basic_world(rl, extra_shapes=(), extra_mats=(), extra_tr=(), light=((-10, 10, -10), (1, 1, 1)), s1=None, s2=None, lights=None):
    """World::basic() (world.rs:34-44,173-198): two concentric spheres, one light; plus optional extra transformed shapes."""
    api = rl.api
    mats = [s1 if s1 is not None else _material(api, color=(0.8, 1.0, 0.6), diffuse=0.7, specular=0.2), s2 if s2 is not None else _material(api)] + list(extra_mats)
    sh = np.zeros(2 + len(extra_shapes), dtype=api.RTC_SHAPE)
    sh["kind"][:2], sh["material"][:2] = api.O_SPHERE, [0, 1]
    for i, (kind, mat) in enumerate(extra_shapes):
        sh["kind"][2 + i], sh["material"][2 + i] = kind, mat
    S = np.diag([0.5, 0.5, 0.5, 1.0])
    tr = [api.rtc_transformed(np.eye(4), api.O_SPHERE, 0), api.rtc_transformed(S, api.O_SPHERE, 1)]
    for i, m in enumerate(extra_tr):
        tr.append(api.rtc_transformed(m, extra_shapes[i][0], 2 + i))
    tr = np.array(tr, dtype=api.RTC_TRANSFORMED)
    objs = np.zeros(len(tr), dtype=api.HREF)
    objs["kind"], objs["index"] = api.O_TRANSFORMED, np.arange(len(tr))
    if lights is None:
        lights = [light]
    lt = np.zeros(len(lights), dtype=api.RTC_LIGHT)
    for i, (pos, inten) in enumerate(lights):
        lt["position"][i], lt["intensity"][i] = pos, inten
    return rl.RtcWorld.from_arrays(np.zeros(0, dtype=api.RTC_TRIANGLE), np.array(mats, dtype=api.RTC_MATERIAL), objs, lt, transformeds=tr, shapes=sh)


def _T(x, y, z):
    m = np.eye(4)
    m[:3, 3] = (x, y, z)
    return m


def _close(c, want):
    return np.abs(np.asarray(c) - np.asarray(want)).max() <= 1e-5  # color::test_utils::assert_colors_approx_equal


def test_world_color_at_known_answers(rl, oracle):
    w = _basic_world(rl)
    assert _ts(oracle, w, (0, 0, -5), (0, 0, 1)) == [4.0, 4.5, 5.5, 6.0]                       # intersect_world_with_ray
    assert tuple(oracle.rtc_color_at(w.desc, (0, 0, -5), (0, 1, 0))) == (0.0, 0.0, 0.0)         # color_when_a_ray_misses
    assert _close(oracle.rtc_color_at(w.desc, (0, 0, -5), (0, 0, 1)), (0.38066, 0.47583, 0.2855))  # color_when_a_ray_hits / shading_an_intersection
    inside = _basic_world(rl, light=((0, 0.25, 0), (1, 1, 1)))
    assert _close(oracle.rtc_color_at(inside.desc, (0, 0, 0), (0, 0, 1)), (0.90498, 0.90498, 0.90498))  # shading_an_intersection_from_the_inside
    dark = _basic_world(rl, lights=[])
    assert tuple(oracle.rtc_color_at(dark.desc, (0, 0, -5), (0, 0, 1))) == (0.0, 0.0, 0.0)      # shading_when_there_are_no_lights
    # color_with_an_intersection_behind_the_ray: both spheres ambient 1, ray from between them looking inwards -> the inner colour
    api = rl.api
    amb = _basic_world(rl, s1=_material(api, color=(0.8, 1.0, 0.6), diffuse=0.7, specular=0.2, ambient=1.0), s2=_material(api, ambient=1.0))
    assert _close(oracle.rtc_color_at(amb.desc, (0, 0, 0.75), (0, 0, -1)), (1.0, 1.0, 1.0))


def test_world_reflection_and_refraction_known_answers(rl, oracle):
    api = rl.api
    ray = ((0, 0, -3), (0, -SQ2 / 2.0, SQ2 / 2.0))
    refl = _basic_world(rl, extra_shapes=[(api.O_PLANE, 2)], extra_mats=[_material(api, reflectivity=0.5)], extra_tr=[_T(0, -1, 0)])
    assert _close(oracle.rtc_color_at(refl.desc, *ray), (0.87675, 0.92434, 0.82917))            # shade_hit_with_a_reflective_material
    red_ball = _material(api, color=(1, 0, 0), ambient=0.5)
    transp = _basic_world(rl, extra_shapes=[(api.O_PLANE, 2), (api.O_SPHERE, 3)], extra_mats=[_material(api, transparency=0.5, refractive_index=1.5), red_ball],
                          extra_tr=[_T(0, -1, 0), _T(0, -3.5, -0.5)])
    assert _close(oracle.rtc_color_at(transp.desc, *ray), (1.12546, 0.68642, 0.68642))          # shade_hit_with_a_transparent_material
    both = _basic_world(rl, extra_shapes=[(api.O_PLANE, 2), (api.O_SPHERE, 3)],
                        extra_mats=[_material(api, transparency=0.5, refractive_index=1.5, reflectivity=0.5), red_ball], extra_tr=[_T(0, -1, 0), _T(0, -3.5, -0.5)])
    assert _close(oracle.rtc_color_at(both.desc, *ray), (1.11500, 0.69643, 0.69243))            # shade_hit_with_a_reflective_transparent_material
    # color_at_with_mutually_reflective_surfaces: two facing mirrors must terminate (max_reflection_depth)
    sh = np.zeros(2, dtype=api.RTC_SHAPE)
    sh["kind"], sh["material"] = api.O_PLANE, 0
    tr = np.array([api.rtc_transformed(_T(0, -1, 0), api.O_PLANE, 0), api.rtc_transformed(_T(0, 1, 0), api.O_PLANE, 1)], dtype=api.RTC_TRANSFORMED)
    objs = np.zeros(2, dtype=api.HREF)
    objs["kind"], objs["index"] = api.O_TRANSFORMED, [0, 1]
    lt = np.zeros(1, dtype=api.RTC_LIGHT)
    lt["position"], lt["intensity"] = (0, 0, 0), (1, 1, 1)
    mirrors = rl.RtcWorld.from_arrays(np.zeros(0, dtype=api.RTC_TRIANGLE), np.array([_material(api, reflectivity=1.0)], dtype=api.RTC_MATERIAL), objs, lt,
                                      transformeds=tr, shapes=sh)
    assert np.isfinite(oracle.rtc_color_at(mirrors.desc, (0, 0, 0), (0, 1, 0))).all()
