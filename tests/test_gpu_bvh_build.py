"""GPU tier: Bvh::new on the device (rl_bvh_build) against the host mirror's recursive build
(rendering-learning_amd/host/rtiow_host.hpp Bvh, which restates bvh.rs:22-60): every node record identical — boxes bit for
bit, same children in the same order — including worlds where many sort keys are equal (stable order)."""
import ctypes as C
import time

import numpy as np
import pytest

pytestmark = pytest.mark.gpu


@pytest.mark.parametrize("n,ties", [(1, 0), (2, 0), (3, 0), (4, 0), (5, 0), (7, 1), (64, 0), (100, 1), (1000, 0), (1000, 1), (4097, 0), (50000, 1), (200000, 0)])
def test_device_bvh_equals_host_bvh(rl, n, ties):
    rl.init(0)
    L = rl.api.host_lib()
    L.rlh_bvh_build_compare.restype = C.c_int64
    L.rlh_bvh_build_compare.argtypes = [C.c_uint32, C.c_uint64, C.c_int]
    r = L.rlh_bvh_build_compare(n, 11 + n, ties)
    assert r == 0, (r, L.rlh_last_error().decode())


def _random_boxes(n, seed, ties):
    rng = np.random.default_rng(seed)
    c = rng.uniform(-20, 20, (n, 3))
    if ties:  # snap to a coarse grid: many equal sort keys, so the (stable) order among them matters
        c = np.floor(c / 4.0) * 4.0
    r = rng.uniform(0.05, 1.0, (n, 3)) if not ties else np.full((n, 3), 0.5)
    boxes = np.empty((n, 6))
    boxes[:, 0::2], boxes[:, 1::2] = c - r, c + r
    if n > 8:  # a few degenerate (zero-extent, already padded by AABB::new) and huge boxes, and a negative zero key
        boxes[3, 0::2], boxes[3, 1::2] = c[3] - 5e-5, c[3] + 5e-5
        boxes[5] = [-1e30, 1e30, -1.0, 1.0, -0.0, 2.0]
        boxes[6, 4] = -0.0
    return boxes


@pytest.mark.parametrize("n,ties", [(1, 0), (2, 0), (3, 0), (4, 0), (5, 0), (7, 1), (64, 0), (100, 1), (1000, 0), (1000, 1), (4097, 0), (50000, 1), (200000, 0)])
def test_device_bvh_equals_oracle_bvh(rl, oracle, n, ties):
    """rl_bvh_build (HIP) against the ORACLE's restatement of Bvh::new (oracle/rl_oracle.cpp rlo_bvh_build, bvh.rs:22-77): every
    node record byte-identical — boxes bit for bit, same children in the same order, same node numbering."""
    rl.init(0)
    api = rl.api
    L = api.render_lib()
    L.rl_bvh_build.argtypes = [C.c_void_p, C.c_void_p, C.c_uint32, C.c_uint32, C.c_void_p, C.c_uint32, C.c_void_p]
    boxes = _random_boxes(n, 77 + n, ties)
    prims = np.zeros(n, dtype=api.HREF)
    prims["kind"] = np.where(np.arange(n) % 5 == 4, 2, 1)
    prims["index"] = np.arange(n)[::-1]
    base = 17
    want = oracle.bvh_build(boxes, prims, node_base=base)
    got = np.zeros(len(want) + 3, dtype=oracle.BVH_NODE)
    cnt = C.c_uint32()
    assert L.rl_bvh_build(boxes.ctypes.data, prims.ctypes.data, n, base, got.ctypes.data, len(got), C.byref(cnt)) == 0, L.rl_last_error()
    assert cnt.value == len(want)
    assert got[:cnt.value].tobytes() == want.tobytes()


def test_bvh_build_rejects_bad_input(rl):
    rl.init(0)
    api = rl.api
    L = api.render_lib()
    L.rl_bvh_build.argtypes = [C.c_void_p, C.c_void_p, C.c_uint32, C.c_uint32, C.c_void_p, C.c_uint32, C.c_void_p]
    boxes = np.zeros((3, 6))
    boxes[:, 1::2] = 1.0
    prims = np.zeros(3, dtype=api.HREF)
    prims["kind"], prims["index"] = 1, np.arange(3)
    nodes = np.zeros(8 * 64, dtype=np.uint8)
    cnt = C.c_uint32()
    assert L.rl_bvh_build(boxes.ctypes.data, prims.ctypes.data, 0, 0, nodes.ctypes.data, 8, C.byref(cnt)) == -1   # Bvh::new panics on an empty list
    assert L.rl_bvh_build(boxes.ctypes.data, prims.ctypes.data, 3, 0, nodes.ctypes.data, 1, C.byref(cnt)) == -1 and cnt.value == 3  # capacity
    boxes[1, 2] = np.nan
    assert L.rl_bvh_build(boxes.ctypes.data, prims.ctypes.data, 3, 0, nodes.ctypes.data, 8, C.byref(cnt)) == -1
    boxes[1, 2] = 0.0
    assert L.rl_bvh_build(boxes.ctypes.data, prims.ctypes.data, 3, 0, nodes.ctypes.data, 8, C.byref(cnt)) == 0 and cnt.value == 3


def test_stress_scene_with_device_built_bvhs_renders_identically(rl, golden):
    """cfg-5 generator, reduced: spheres + subdivided mesh under an instance; host-built vs device-built trees give the same
    program up to primitive numbering: identical pixels and identical counters."""
    import gzip  # noqa: F401
    from PIL import Image
    import os
    tex = np.asarray(Image.open(os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden", "spot_texture.png")).convert("RGB"))
    obj = golden("spot_triangulated.obj.gz")
    t0 = time.perf_counter()
    host = rl.World.stress_scene(40, 0, obj, tex)
    t1 = time.perf_counter()
    dev = rl.World.stress_scene(40, 0, obj, tex, device_bvh=True)
    t2 = time.perf_counter()
    print(f"host build {t1 - t0:.2f} s, device build {t2 - t1:.2f} s")
    p = host.params
    p.image_width, p.samples_per_pixel, p.max_depth = 160, 4, 12
    cam = rl.Camera(p)
    a, b = {}, {}
    ia = cam.render(host, stats=a).data
    ib = cam.render(dev, stats=b).data
    assert np.array_equal(ia, ib)
    for k in ("rays", "node_tests", "sphere_tests", "planar_tests", "instance_enters", "rng_words", "flagged"):
        assert a[k] == b[k], k
