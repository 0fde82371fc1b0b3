"""GPU tier: seeded DIFFERENTIAL FUZZ of the fast traversal for GENERAL scenes (csrc/rl_rtiow_fastgen.h, rl_fast_bvh.cpp: world-space SAH
trees folded four wide, one per program segment; constant media between the segments behind box nodes, with the one-pass boundary
evaluation for planar lists and spheres; unbounded Planes as stages of their own; the tree's top in LDS) against the counting
(reference-order) kernel: frames must agree BIT FOR BIT, ray counts and panic-site counts exactly.  The worlds are random compositions of
everything the scene vocabulary offers — spheres (moving too), quads, triangles (with vertex normals), unbounded planes, lists and Bvhs
nested in each other, Translate / rotate / scale instances (nested, shared), constant media whose boundaries are spheres, boxes of quads,
tetrahedra, Bvhs of spheres or plane slabs, every material — with touching, overlapping and coincident parts thrown in, because the exactness
of the fast path rests on its tie bands and paddings.  A subset of the worlds is also rendered by the oracle (counters equal, colours within
1e-9 relative: the medium's log() is the only libm call that differs)."""
import os

import numpy as np
import pytest

pytestmark = pytest.mark.gpu


def _random_scene(rng, family):
    """Returns a function(builder) -> root id; every random number is drawn from `rng` up front so that the scene is a pure function of the seed."""
    R = np.random.default_rng(int(rng.integers(0, 1 << 62)))

    def scene(b):
        tex = [b.solid(tuple(R.uniform(0.1, 0.95, 3))) for _ in range(3)]
        tex.append(b.checker(float(R.uniform(0.3, 2.0)), tex[0], tex[1]))
        mats = [b.lambertian(t) for t in tex] + [b.metal(tuple(R.uniform(0.5, 0.95, 3)), float(R.choice([0.0, 0.2]))), b.dielectric(1.5),
                                                 b.diffuse_light(b.solid(tuple(R.uniform(2, 6, 3))))]
        flat = b.flat()
        mat = lambda: mats[int(R.integers(0, len(mats)))]
        span = 4.0

        def point(scale=span):
            return tuple(R.uniform(-scale, scale, 3))

        def sphere(m=None):
            c, r = point(), float(R.uniform(0.15, 1.2))
            mv = R.random() < 0.2
            return b.sphere(c, r, m if m is not None else mat(), center2=tuple(np.add(c, R.uniform(-0.3, 0.3, 3))) if mv else None)

        def quad(m=None):
            return b.quad(point(), tuple(R.uniform(-2, 2, 3)), tuple(R.uniform(-2, 2, 3)), m if m is not None else mat())

        def triangle(m=None):
            q = np.array(point())
            pts = [q, q + R.uniform(-2, 2, 3), q + R.uniform(-2, 2, 3)]
            if R.random() < 0.4:  # vertex normals that point roughly the same way (a vanishing interpolated normal is refused by the builder)
                n0 = R.normal(size=3)
                n0 /= np.linalg.norm(n0)
                normals = [n0 + R.uniform(-0.3, 0.3, 3) for _ in range(3)]
                return b.triangle_from_model(pts, m if m is not None else mat(), normals=normals)
            return b.triangle(tuple(pts[0]), tuple(pts[1] - pts[0]), tuple(pts[2] - pts[0]), m if m is not None else mat())

        def box(lo, hi, m):
            (x0, y0, z0), (x1, y1, z1) = lo, hi
            dx, dy, dz = (x1 - x0, 0, 0), (0, y1 - y0, 0), (0, 0, z1 - z0)
            return [b.quad((x0, y0, z1), dx, dy, m), b.quad((x1, y0, z1), (0, 0, -(z1 - z0)), dy, m), b.quad((x1, y0, z0), (-(x1 - x0), 0, 0), dy, m),
                    b.quad((x0, y0, z0), dz, dy, m), b.quad((x0, y1, z1), dx, (0, 0, -(z1 - z0)), m), b.quad((x0, y0, z0), dx, dz, m)]

        def primitive():
            k = R.random()
            return sphere() if k < 0.45 else quad() if k < 0.75 else triangle()

        def instance(obj):
            k = int(R.integers(0, 5))
            if k == 0:
                return b.translate(obj, tuple(R.uniform(-2, 2, 3)))
            if k == 1:
                return b.rotate_x(obj, float(R.uniform(-80, 80)))
            if k == 2:
                return b.rotate_y(obj, float(R.uniform(-170, 170)))
            if k == 3:
                return b.rotate_z(obj, float(R.uniform(-80, 80)))
            return b.scale(obj, float(R.uniform(0.4, 1.8)))

        def group(depth):
            n = int(R.integers(1, 7))
            parts = []
            for _ in range(n):
                if depth < 2 and R.random() < 0.25:
                    parts.append(group(depth + 1))
                else:
                    parts.append(primitive())
            if R.random() < 0.3 and parts:  # coincident / touching parts: the tie band must hand over
                parts.append(parts[int(R.integers(0, len(parts)))])
            g = b.bvh(parts) if R.random() < 0.6 else b.list(parts)
            for _ in range(2):  # (the flattener refuses instances nested deeper than 8: three group levels x 2 + a boundary's 2)
                if R.random() < 0.4:
                    g = instance(g)
            return g

        def medium():
            k = int(R.integers(0, 5 if family == "planes" else 4))
            if k == 0:
                bd = b.sphere(point(3.0), float(R.uniform(0.5, 1.8)), flat)
            elif k == 1:
                lo = np.array(point(3.0))
                bd = b.list(box(tuple(lo), tuple(lo + R.uniform(0.5, 2.5, 3)), flat))
            elif k == 2:
                q = np.array(point(3.0))
                p = [q, q + (1.5, 0, 0), q + (0.7, 0, 1.4), q + (0.7, 1.5, 0.5)]
                tri = lambda i, j, kk: b.triangle(tuple(p[i]), tuple(p[j] - p[i]), tuple(p[kk] - p[i]), flat)
                bd = b.list([tri(0, 1, 2), tri(0, 1, 3), tri(1, 2, 3), tri(2, 0, 3)])
            elif k == 3:
                c = np.array(point(3.0))
                bd = b.bvh([b.sphere(tuple(c + (i * 0.7, 0, 0)), 0.6, flat) for i in range(3)])
            else:  # fog between two parallel planes
                y = float(R.uniform(-1, 1))
                bd = b.list([b.plane((0, y, 0), (1, 0, 0), (0, 0, 1), flat), b.plane((0, y + float(R.uniform(0.2, 1.0)), 0), (1, 0, 0), (0, 0, 1), flat)])
            for _ in range(2):
                if R.random() < 0.35:
                    bd = instance(bd)
            return b.constant_medium(bd, float(R.uniform(0.05, 1.5)), b.isotropic(tex[int(R.integers(0, 3))]))

        top = [group(0) for _ in range(int(R.integers(1, 4)))]
        if family in ("media", "planes", "mixed"):
            for _ in range(int(R.integers(1, 4))):
                top.insert(int(R.integers(0, len(top) + 1)), medium())
        if family in ("planes", "mixed"):
            for _ in range(int(R.integers(1, 3))):
                n = R.normal(size=3)
                u = np.cross(n, (0.3, 1.0, 0.2))
                pl = b.plane(tuple(-n / np.linalg.norm(n) * R.uniform(2, 5)), tuple(u), tuple(np.cross(n, u)), mats[int(R.integers(0, 4))])
                if R.random() < 0.3:
                    pl = instance(pl)
                top.insert(int(R.integers(0, len(top) + 1)), pl)
        if family == "mixed" and R.random() < 0.5:  # a medium INSIDE a Bvh next to world parts
            top.append(b.bvh([medium(), primitive(), primitive()]))
        return b.bvh(top) if (family in ("instances", "media") and R.random() < 0.5) else b.list(top)
    return scene


@pytest.mark.parametrize("family", ["instances", "media", "planes", "mixed"])
def test_fast_general_kernel_equals_the_counting_kernel_on_random_compositions(rl, oracle, family):
    import torch
    api = rl.api
    rng = np.random.default_rng({"instances": 101, "media": 202, "planes": 303, "mixed": 404}[family])
    dev = torch.device("cuda", 0)
    stream = torch.cuda.current_stream(dev).cuda_stream
    out16 = (api.C.c_uint64 * 16)()
    api.render_lib().rl_debug_host_structures.argtypes = [api.C.c_void_p, api.C.c_void_p]
    fast_structures = slow = rays = media = 0
    cases = int(os.environ.get("RL_FUZZ_CASES", "40"))  # (a one-off deep run: RL_FUZZ_CASES=600 RL_FUZZ_WIDTH=150)
    for case in range(cases):
        world = rl.World.build(_random_scene(rng, family))
        frm = tuple(rng.uniform(-1, 1, 3) + (0.0, 1.0, 11.0))
        p = rl.CameraParams(aspect_ratio=1.5, image_width=int(os.environ.get("RL_FUZZ_WIDTH", "60")), samples_per_pixel=6, max_depth=int(rng.choice([3, 12])), vfov=50.0, lookfrom=frm,
                            lookat=(0.0, 0.0, 0.0), defocus_angle=float(rng.choice([0.0, 0.4])), focus_dist=11.0,
                            background=(0.5, 0.6, 0.9), seed=int(rng.integers(0, 1 << 30)))
        cam = rl.Camera(p)
        gs = {}
        counting = cam.render(world, stats=gs, allow_degenerate=True).data
        assert api.render_lib().rl_debug_host_structures(world.desc, out16) == 0
        assert out16[5] == 0 and out16[6] == 0 and out16[7] == 0, (family, case, list(out16))  # no duplicate / missing items, no box violations
        fast_structures += int((out16[0] >> 1) & 1)
        media += int(out16[14])
        buf = torch.full((cam.c.image_height, cam.c.image_width, 3), float("nan"), dtype=torch.float64, device=dev)
        cam.render_device(world, buf.data_ptr(), stream=stream)
        st = api.render_status(world, allow_degenerate=True)
        fast = buf.cpu().numpy()
        assert np.array_equal(fast.view(np.uint64), counting.view(np.uint64)), (family, case, int((fast != counting).any(axis=2).sum()))
        assert st["rays"] == gs["rays"] and st["flagged"] == gs["flagged"], (family, case)
        slow += st["slow_traces"]
        rays += gs["rays"]
        if case % 5 == 0:  # the oracle on every fifth world: the counting kernel IS the reference's fold
            cs = {}
            cpu = oracle.rtiow_render(world.desc, cam.c, stats=cs)
            for k in ("rays", "node_tests", "sphere_tests", "planar_tests", "instance_enters", "rng_words", "flagged"):
                assert gs[k] == cs[k], (family, case, k, gs[k], cs[k])
            assert np.abs(counting - cpu).max() <= 1e-9 * max(1.0, np.abs(cpu).max()), (family, case)
    assert fast_structures >= 30, fast_structures  # the fuzz must exercise the fast structure, not only its fallbacks
    assert slow > 0 and slow < rays // 4            # coincident parts: the tie band fires — and stays the exception
    if family != "instances":
        assert media >= 40
