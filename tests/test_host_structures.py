"""CPU tier: the traversal structures a scene compiles to on the HOST (csrc/rl_fast_bvh.cpp: the sphere kernel's ordered SAH tree, the
general kernel's world-space tree folded four wide) checked without a device through rl_debug_host_structures: every primitive
occurrence is reachable exactly once, every box contains what lies below it, depth bounds hold, and the eligibility rules send the right
scenes to the reference-order kernels."""
import ctypes as C
import os

import numpy as np
import pytest


def _check(rl, world):
    L = rl.api.render_lib()
    L.rl_debug_host_structures.argtypes = [C.c_void_p, C.c_void_p]
    out = (C.c_uint64 * 16)()
    rc = L.rl_debug_host_structures(world.desc, out)
    assert rc == 0, L.rl_last_error().decode()
    keys = ("flags", "items", "binary_nodes", "quad_nodes", "leaves", "dups", "missing", "box_violations", "quad_depth",
            "s_inner", "s_leaves", "s_bad_reach", "s_box_violations", "s_depth", "media", "media_shapes")
    return dict(zip(keys, list(out)))


def _tex():
    from PIL import Image
    root = os.path.dirname(os.path.abspath(__file__))
    return np.asarray(Image.open(os.path.join(root, "golden", "spot_texture.png")).convert("RGB"))


def test_sphere_tree_of_the_baseline_scene(rl):
    world = rl.World.bouncing_spheres(1)
    r = _check(rl, world)
    n = world.counts()["spheres"]
    assert r["flags"] == 1 and n == 488
    assert r["s_inner"] == n - 1 and r["s_leaves"] == n and r["s_bad_reach"] == 0 and r["s_box_violations"] == 0
    assert 9 <= r["s_depth"] <= 15  # FAST_MAX_DEPTH = 14 levels of inner nodes + the leaf
    r = _check(rl, rl.World.golden_test_scene())
    assert r["flags"] == 1 and r["s_leaves"] == 5 and r["s_bad_reach"] == 0 and r["s_box_violations"] == 0


def test_general_structure_reaches_every_occurrence_once(rl, golden):
    tex = _tex()
    scenes = {
        "cow": rl.World.cow_scene(golden("spot_triangulated.obj.gz"), tex),
        "cornell_box": rl.World.example_scene("cornell_box"),
        "teapot": rl.World.example_scene("teapot", obj_text=golden("teapot-low.obj")),
        "stress": rl.World.stress_scene(40, 1, golden("spot_triangulated.obj.gz"), tex),
        "quads": rl.World.example_scene("quads"),
        "earth": rl.World.earth_scene(tex[::8, ::8]),
    }
    for name, world in scenes.items():
        r = _check(rl, world)
        c = world.counts()
        assert r["flags"] == 2, (name, r)
        assert r["leaves"] == r["items"] and r["dups"] == 0 and r["missing"] == 0 and r["box_violations"] == 0, (name, r)
        assert r["items"] >= c["spheres"] + c["planars"], (name, r, c)  # one item per occurrence (>= one per primitive)
        if r["items"] > 1:
            assert r["binary_nodes"] == r["items"] - 1 and r["quad_nodes"] <= r["binary_nodes"], (name, r)
            # folding every second level: a node holds 2 .. 4 children, so the four-wide tree has at least a third as many nodes
            assert 3 * r["quad_nodes"] >= r["binary_nodes"] and r["quad_depth"] <= 41, (name, r)
    assert _check(rl, scenes["cow"])["items"] == scenes["cow"].counts()["planars"]  # 6 quads + 5856 triangles, one occurrence each


def test_instanced_twice_counts_occurrences_not_primitives(rl):
    def build(b):
        mat = b.lambertian(b.solid((0.6, 0.6, 0.7)))
        mesh = b.bvh([b.triangle((0, 0, 0), (1, 0, 0), (0, 1, 0), mat), b.quad((0, 0, 0), (0, 0, 1), (0, 1, 0), mat), b.sphere((0.4, 0.4, 0.4), 0.25, mat)])
        return b.bvh([b.translate(mesh, (-1.2, 0, -3)), b.translate(b.rotate_y(mesh, 70.0), (0.6, -0.2, -3.5)), b.sphere((0, -100.5, -3), 100, mat)])
    r = _check(rl, rl.World.build(build))
    assert r["flags"] == 2 and r["items"] == 7 and r["leaves"] == 7 and r["dups"] == 0 and r["missing"] == 0 and r["box_violations"] == 0


def test_scenes_that_must_stay_on_the_reference_order_kernels(rl):
    # an unbounded Plane has no box: since round 3 it is a stage of its own next to the tree of the four bounded parts (every ray tests it)
    flat = _check(rl, rl.World.example_scene("flat_world"))
    assert flat["flags"] == 2 and flat["items"] == 5 and flat["leaves"] == 5 and flat["missing"] == 0 and flat["box_violations"] == 0
    # constant media (round 3): the items are cut into program segments at the media — the six Cornell quads come before both smoke boxes,
    # whose own boundary quads are not world primitives — and every segment's tree is checked like the single one of a scene without media
    smoke = _check(rl, rl.World.example_scene("cornell_smoke"))
    assert smoke["flags"] == 2 and smoke["items"] == 6 and smoke["leaves"] == 6 and smoke["dups"] == 0 and smoke["missing"] == 0 and smoke["box_violations"] == 0
    # ... both media sit behind a box node, and their boundaries (Translate(RotateY([Quad; 6]))) are recognised as planar lists (shape 1)
    assert smoke["media"] == 2 and smoke["media_shapes"] == 0x1111
    final = _check(rl, rl.World.example_scene("final_scene", rgb8=_tex()))  # two sphere boundaries (shape 2)
    assert final["flags"] == 2 and final["media"] == 2 and final["media_shapes"] == 0x1212 and final["box_violations"] == 0
    import test_constant_medium as tcm  # sphere, moving sphere under a Translate, rotated box, tetrahedron, 18 quads, a Bvh of spheres (general), one quad
    shapes = _check(rl, rl.World.build(tcm._boundary_shapes_scene))
    assert shapes["flags"] == 2 and shapes["media"] == 7 and shapes["media_shapes"] == 0x11101111111212
    assert _check(rl, rl.World.example_scene("checkered_spheres"))["flags"] == 1

    def vanishing(b):  # a smooth triangle whose interpolated normal passes through zero
        mat = b.lambertian(b.solid((0.7, 0.6, 0.5)))
        return b.bvh([b.triangle_from_model([[-1, 0, -3], [1, 0, -3], [0, 1, -3]], mat, normals=[[0, 0, 1], [0.6, 0, 0.8], [0, 0, -1]]),
                      b.quad((-3, -1.7, -5), (6, 0, 0), (0, 0, 5), mat)])
    assert _check(rl, rl.World.build(vanishing))["flags"] == 0

    def apart(b):  # vertex normals that merely point apart (shortest interpolated normal 0.6): fine
        mat = b.lambertian(b.solid((0.7, 0.6, 0.5)))
        return b.bvh([b.triangle_from_model([[-1, 0, -3], [1, 0, -3], [0, 1, -3]], mat, normals=[[-0.8, 0, 0.6], [0.8, 0, 0.6], [0, 0.9, 0.45]]),
                      b.quad((-3, -1.7, -5), (6, 0, 0), (0, 0, 5), mat)])
    assert _check(rl, rl.World.build(apart))["flags"] == 2

    def radius_zero(b):
        mat = b.lambertian(b.solid((0.5, 0.5, 0.5)))
        return b.list([b.sphere((0, 0, -1), 0.0, mat), b.sphere((0, -100.5, -1), 100, mat)])
    assert _check(rl, rl.World.build(radius_zero))["flags"] == 0
