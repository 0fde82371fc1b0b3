"""GPU tier: seeded DIFFERENTIAL FUZZ of the default product path for sphere worlds — the fast traversal of rtiow_wave_kernel<1024, 4, false>
and the cooperative one-wave-per-pixel kernel (csrc/rl_rtiow_wave.h, rl_rtiow_coop.h, rl_fast_bvh.cpp) — against the counting
(reference-order) kernel: frames must agree BIT FOR BIT, ray and panic-site counts exactly.  The exactness of the fast paths rests on error
margins (tie band, guard padding, pole / grazing checks: rl_rtiow_wave.h:199-263); the worlds here are built to lean on those margins:
radius ratios up to 1e6, near-coincident and nested spheres (shells 1e-9 ... 1e-3 apart), spheres touching, cameras at 0.5 ... 0.999 of the
guard frame's reach and just outside it, coordinates from 1e-6 to 1e6, moving spheres, every material.  96 worlds, a few seconds in total.
(The per-ray form of the same comparison is the verification build: make verify + tools/verify_fastg.py <spp> spheres | coop.)"""
import numpy as np
import pytest

pytestmark = pytest.mark.gpu


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


def _world(api, rng, kind):
    """One random world of the given family; returns (spheres, lookfrom, lookat, vfov)."""
    scale = 10.0 ** rng.uniform(-6, 6) if kind == "scaled" else 1.0
    if kind == "ratios":  # radius ratios up to 1e6: pebbles on and around boulders
        n = int(rng.integers(3, 40))
        # half of the worlds stay below the ratio at which the from_normalized assert (vec3.rs:219) becomes reachable and the scene
        # therefore gets NO fast structure (GuardFrame::normals_safe: extent / radius <~ 2.6e4); the other half go up to 1e6 (fallback path)
        tame = rng.random() < 0.5
        big = 10.0 ** (rng.uniform(1, 1.5) if tame else rng.uniform(1, 3))
        c = rng.uniform(-2, 2, (n, 3))
        r = 10.0 ** (rng.uniform(-2, 0, n) if tame else rng.uniform(-3, 0, n)) * 0.5
        c[0], r[0] = (0.0, -big - 0.5, 0.0), big  # the ground: big / min(r) up to 1e6
        k = n // 2
        d = rng.normal(size=(k, 3))
        d /= np.linalg.norm(d, axis=1, keepdims=True)
        d[:, 1] = np.abs(d[:, 1])
        c[1:1 + k] = c[0] + d * (big + r[1:1 + k])[:, None]  # touching the ground sphere
        frm, to, fov = (0.0, 1.0, 6.0), (0.0, 0.0, 0.0), 45.0
    elif kind == "nested":  # shells 1e-9 ... 1e-3 apart, shared centres, near-coincident neighbours
        n = int(rng.integers(4, 30))
        base = rng.uniform(-1.5, 1.5, (n, 3))
        c = base.copy()
        r = rng.uniform(0.2, 0.8, n)
        for i in range(1, n):
            if rng.random() < 0.7:
                j = int(rng.integers(0, i))
                eps = 10.0 ** rng.uniform(-9, -3)
                c[i] = c[j] + (rng.normal(size=3) * eps if rng.random() < 0.5 else 0.0)
                r[i] = r[j] + eps * rng.choice([-1.0, 1.0, 0.0])
        frm, to, fov = (0.0, 0.5, 5.0), (0.0, 0.0, 0.0), 50.0
    elif kind == "reach":  # a compact cluster seen from 0.5 ... 0.999 of the guard frame's reach (one scene diameter) — and from outside
        n = int(rng.integers(2, 60))
        c = rng.uniform(-1, 1, (n, 3))
        r = rng.uniform(0.05, 0.4, n)
        ext = float(np.linalg.norm((c + r[:, None]).max(0) - (c - r[:, None]).min(0)))
        f = rng.choice([0.5, 0.9, 0.99, 0.999, 1.001, 1.2])
        dirv = rng.normal(size=3)
        dirv /= np.linalg.norm(dirv)
        centre = 0.5 * ((c + r[:, None]).max(0) + (c - r[:, None]).min(0))
        frm, to, fov = tuple(centre + dirv * ext * f), tuple(centre), 40.0
    else:  # "scaled": an ordinary cluster at coordinates 1e-6 ... 1e6 (the margins are relative: nothing may change)
        n = int(rng.integers(2, 80))
        c = rng.uniform(-3, 3, (n, 3)) * scale
        r = rng.uniform(0.05, 0.9, n) * scale
        frm, to, fov = (0.0, 1.0 * scale, 9.0 * scale), (0.0, 0.0, 0.0), 50.0
    sph = np.zeros(len(r), dtype=api.SPHERE)
    sph["center0"], sph["radius"] = c, r
    mv = rng.random(len(r)) < 0.2
    sph["moving"] = mv
    sph["center1"] = c + rng.uniform(0, 0.3, c.shape) * r[:, None] * mv[:, None]
    sph["material"] = rng.integers(0, 5, len(r))
    return sph, frm, to, fov


@pytest.mark.parametrize("kind", ["ratios", "nested", "reach", "scaled"])
def test_fast_and_cooperative_kernels_equal_the_counting_kernel_on_adversarial_worlds(rl, kind):
    import torch
    api = rl.api
    tex, mats = _mats(api)
    rng = np.random.default_rng({"ratios": 11, "nested": 22, "reach": 33, "scaled": 44}[kind])
    dev = torch.device("cuda", 0)
    stream = torch.cuda.current_stream(dev).cuda_stream
    fast_structures = slow = rays = 0
    for case in range(24):
        sph, frm, to, fov = _world(api, rng, kind)
        world = rl.World.from_spheres(sph, mats, tex, bool(rng.integers(0, 2)))
        p = rl.CameraParams(aspect_ratio=1.5, image_width=48, samples_per_pixel=4, max_depth=10, vfov=fov, lookfrom=frm, lookat=to,
                            defocus_angle=float(rng.choice([0.0, 0.5])), focus_dist=float(np.linalg.norm(np.subtract(frm, to))),
                            background=(0.5, 0.6, 0.9), seed=int(rng.integers(0, 1 << 30)))
        cam = rl.Camera(p)
        gs = {}
        counting = cam.render(world, stats=gs, allow_degenerate=True).data
        out16 = (api.C.c_uint64 * 16)()
        api.render_lib().rl_debug_host_structures.argtypes = [api.C.c_void_p, api.C.c_void_p]
        assert api.render_lib().rl_debug_host_structures(world.desc, out16) == 0
        fast_structures += int(out16[0] & 1)
        frames = {}
        try:
            for v in (1029, 1033):  # the fast wave-scheduled kernel / the cooperative kernel (each falls back by itself when a scene does not qualify)
                api.set_rtiow_variant(v)
                buf = torch.full((cam.c.image_height, cam.c.image_width, 3), float("nan"), dtype=torch.float64, device=dev)
                cam.render_device(world, buf.data_ptr(), stream=stream)
                st = api.render_status(world, allow_degenerate=True)
                frames[v] = (buf.cpu().numpy(), st)
        finally:
            api.set_rtiow_variant(0)
        for v, (img, st) in frames.items():
            assert np.array_equal(img.view(np.uint64), counting.view(np.uint64)), (kind, case, v, int((img != counting).any(axis=2).sum()))
            assert st["rays"] == gs["rays"] and st["flagged"] == gs["flagged"], (kind, case, v)
        slow += frames[1029][1]["slow_traces"]
        rays += gs["rays"]
    assert fast_structures >= (6 if kind == "ratios" else 12), fast_structures  # the fuzz must exercise the fast structure, not only its fallbacks
    if kind == "nested":
        assert slow > 0  # near-coincident shells: the tie band must fire
