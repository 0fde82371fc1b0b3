"""GPU tier, opt-in: the WAVEFRONT form of the general fast traversal (csrc/experimental/rl_rtiow_wfg.h) — a logic kernel, a
slow-trace kernel and a traversal-only kernel per pass, one ray per pixel per pass.  Measured slower than the megakernel and therefore only in
librl_render_exp.so (make -C rendering-learning_amd/csrc exp; RL_RENDER_LIB=.../librl_render_exp.so); it must still render the product's bits."""
import gzip
import os

import numpy as np
import pytest

pytestmark = pytest.mark.gpu
GOLDEN = os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden")


def _device_frame(rl, cam, world):
    import torch
    buf = torch.full((cam.c.image_height, cam.c.image_width, 3), float("nan"), dtype=torch.float64, device="cuda:0")
    cam.render_device(world, buf.data_ptr(), stream=torch.cuda.current_stream().cuda_stream)
    st = rl.api.render_status(world)
    return buf.cpu().numpy(), st


@pytest.mark.parametrize("scene", ["cow", "stress"])
def test_wavefront_form_renders_the_megakernels_bits(rl, scene):
    if not rl.api.has_experimental():
        pytest.skip("product library loaded (the wavefront form lives in librl_render_exp.so; set RL_RENDER_LIB)")
    from PIL import Image
    tex = np.asarray(Image.open(os.path.join(GOLDEN, "spot_texture.png")).convert("RGB"))
    obj = gzip.open(os.path.join(GOLDEN, "spot_triangulated.obj.gz"), "rb").read()
    if scene == "cow":
        world = rl.World.cow_scene(obj, tex)
        p = world.params
        p.image_width, p.samples_per_pixel = 150, 6
    else:  # reduced cfg-5 generator: rays that start inside the ground sphere, order-sensitive rays (the slow queue), moving spheres
        world = rl.World.stress_scene(60, 1, obj, tex)
        p = world.params
        p.image_width, p.samples_per_pixel = 200, 6
    cam = rl.Camera(p)
    gs = {}
    counting = cam.render(world, stats=gs).data
    frames = {}
    try:
        for v in (1031, 1035):  # megakernel / wavefront form of the same fast traversal
            rl.api.set_rtiow_variant(v)
            frames[v] = _device_frame(rl, cam, world)
    finally:
        rl.api.set_rtiow_variant(0)
    assert np.array_equal(frames[1031][0], frames[1035][0])
    assert np.array_equal(frames[1035][0], counting)
    assert frames[1035][1]["rays"] == gs["rays"] == frames[1031][1]["rays"]
    assert frames[1035][1]["slow_traces"] == frames[1031][1]["slow_traces"]
