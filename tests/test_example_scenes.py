"""The reference's remaining example scenes (examples/checkered_spheres.rs, quads.rs, flat_world.rs, cornell_box.rs, cornell_smoke.rs,
teapot.rs, final_scene.rs) through the host mirror (rendering-learning_amd/host/scenes.hpp): the callers on the scene-building side of
the hot path.  CPU tier: the scenes flatten to what the examples describe and the oracle renders them.  GPU tier: timed kernel =
counting kernel bit for bit, counters = the oracle's, colours within 1e-9 (bar: 1e-4).  No reference output exists for any of them
(the examples print to stdout and the repository holds no image of theirs): parity is pinned through the oracle only; the constant
media of cornell_smoke / final_scene are the deterministic variant (include/rl_render.h rl_medium)."""
import os

import numpy as np
import pytest

COUNTERS = ("rays", "node_tests", "sphere_tests", "planar_tests", "instance_enters", "rng_words", "flagged")


def _texture():
    from PIL import Image
    root = os.path.dirname(os.path.abspath(__file__))
    return np.asarray(Image.open(os.path.join(root, "golden", "spot_texture.png")).convert("RGB"))[::8, ::8]


def _scenes(rl, golden):
    return [("checkered_spheres", {}), ("quads", {}), ("flat_world", {}), ("cornell_box", {}), ("cornell_smoke", {}),
            ("teapot", dict(obj_text=golden("teapot-low.obj"))), ("final_scene", dict(rgb8=_texture()))]


def _small(p, name):
    p.image_width, p.samples_per_pixel = 40, 4
    p.max_depth = min(p.max_depth, 12)
    return p


def test_example_scenes_have_the_examples_contents(rl, golden):
    want = {  # (spheres, planars, media) as the example files list them; make_box = 6 quads
        "checkered_spheres": (2, 0, 0), "quads": (0, 5, 0), "flat_world": (0, 5, 0), "cornell_box": (0, 6 + 12, 0), "cornell_smoke": (0, 6 + 12, 2),
        "final_scene": (1000 + 8, 400 * 6 + 1, 2)}
    for name, kw in _scenes(rl, golden):
        world = rl.World.example_scene(name, **kw)
        counts = world.counts()
        if name in want:
            assert (counts["spheres"], counts["planars"], counts["media"]) == want[name], (name, counts)
        else:
            assert counts["spheres"] == 1 and counts["planars"] == 240  # teapot-low.obj: the 240 triangles of RTC's test_obj_scene
    p = rl.World.example_scene("final_scene", rgb8=_texture()).params
    assert (p.image_width, p.samples_per_pixel, p.max_depth, p.vfov) == (400, 250, 4, 40.0)  # the example's "dev" parameters
    p = rl.World.example_scene("cornell_box").params
    assert (p.image_width, p.samples_per_pixel, p.max_depth, tuple(p.lookfrom)) == (600, 200, 50, (278.0, 278.0, -800.0))


@pytest.mark.parametrize("name", ["checkered_spheres", "quads", "flat_world", "cornell_box", "cornell_smoke", "final_scene"])
def test_oracle_renders_the_example_scenes(rl, oracle, golden, name):
    kw = dict(_scenes(rl, golden))[name]
    world = rl.World.example_scene(name, **kw)
    cam = rl.Camera(_small(world.params, name))
    st = {}
    img = oracle.rtiow_render(world.desc, cam.c, stats=st)
    assert np.isfinite(img).all() and st["flagged"] == 0 and st["rays"] >= 40 * 40 * 4
    if name != "quads":
        assert img.max() > 0.0
    if name in ("cornell_smoke", "final_scene"):  # the media are sampled: rays inside them draw their free path from the pixel's stream
        world2 = rl.World.example_scene("cornell_box") if name == "cornell_smoke" else None
        if world2 is not None:
            st2 = {}
            oracle.rtiow_render(world2.desc, cam.c, stats=st2)
            assert st2["rng_words"] != st["rng_words"]


@pytest.mark.gpu
@pytest.mark.parametrize("name", ["checkered_spheres", "quads", "flat_world", "cornell_box", "cornell_smoke", "teapot", "final_scene"])
def test_gpu_example_scenes_equal_oracle(rl, oracle, golden, name):
    import torch
    kw = dict(_scenes(rl, golden))[name]
    world = rl.World.example_scene(name, **kw)
    cam = rl.Camera(_small(world.params, name))
    gs, cs = {}, {}
    counting = cam.render(world, stats=gs).data
    cpu = oracle.rtiow_render(world.desc, cam.c, stats=cs)
    for k in COUNTERS:
        assert gs[k] == cs[k], (name, k, gs[k], cs[k])
    assert np.abs(counting - cpu).max() <= 1e-9 * max(1.0, np.abs(cpu).max()), name
    dev = torch.device("cuda", 0)
    buf = torch.full((cam.c.image_height, cam.c.image_width, 3), float("nan"), dtype=torch.float64, device=dev)
    cam.render_device(world, buf.data_ptr(), stream=torch.cuda.current_stream(dev).cuda_stream)
    st = rl.api.render_status(world)
    assert st["rays"] == gs["rays"] and st["flagged"] == 0
    assert np.array_equal(buf.cpu().numpy(), counting), name  # the timed (counter-free) kernel: same bits
    if name == "final_scene":  # media + Perlin / sphere-UV code: small frames take the one-wave-per-SIMD instantiation (512 registers) — both forms
        L = rl.api.render_lib()
        L.rl_debug_set_fastg_one_wave.argtypes = [rl.api.C.c_int]
        try:
            for mode in (0, 1):
                L.rl_debug_set_fastg_one_wave(mode)
                buf.fill_(float("nan"))
                cam.render_device(world, buf.data_ptr(), stream=torch.cuda.current_stream(dev).cuda_stream)
                st = rl.api.render_status(world)
                assert st["rays"] == gs["rays"] and np.array_equal(buf.cpu().numpy(), counting), mode
        finally:
            L.rl_debug_set_fastg_one_wave(-1)
