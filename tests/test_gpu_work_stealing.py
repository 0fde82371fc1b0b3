"""GPU tier: work stealing on small shards (csrc/rl_rtiow_wave.h STEAL instantiation, csrc/rl_rtiow_coop.h rtiow_steal_loop).  A shard
with about as many pixels as the GPU has lanes is bound by its longest per-pixel sample chains; waves whose lanes have run out of pixels
take pixels over from lanes that are still rendering — at a sample boundary, with the pixel's exact sums and ChaCha word position — and
continue them with all 64 lanes.  Which wave renders which sample must not change a single bit."""
import numpy as np
import pytest

pytestmark = pytest.mark.gpu


def _frame(rl, cam, world, row_first=0, row_step=1):
    import torch
    dev = torch.device("cuda", 0)
    nrows = rl.api.rows_for(cam.c.image_height, row_first, row_step)
    buf = torch.full((nrows, cam.c.image_width, 3), float("nan"), dtype=torch.float64, device=dev)
    cam.render_device(world, buf.data_ptr(), stream=torch.cuda.current_stream(dev).cuda_stream, row_first=row_first, row_step=row_step)
    st = rl.api.render_status(world)
    return buf.cpu().numpy(), st


@pytest.mark.parametrize("width,spp,shard", [(640, 96, 1), (960, 64, 3), (1920, 72, 8)])
def test_work_stealing_renders_the_same_bits(rl, oracle, width, spp, shard):
    world = rl.World.bouncing_spheres(1)
    p = world.params
    p.image_width, p.samples_per_pixel, p.max_depth = width, spp, 50
    cam = rl.Camera(p)
    npix = rl.api.rows_for(cam.c.image_height, 0, shard) * cam.c.image_width
    assert 40960 < npix <= 3 * 256 * 1024  # above the cooperative kernel's small-frame limit, within the stealing rule
    frames = []
    try:
        for fill in (3.0, 0.0, 3.0, 3.0):  # stealing on / off / on / on: the take-overs differ from run to run, the bits must not
            rl.api.set_steal(fill)
            frames.append(_frame(rl, cam, world, 0, shard))
    finally:
        rl.api.set_steal(3.0)
    ref, st_ref = frames[1]
    for img, st in frames:
        assert np.array_equal(img, ref) and st["rays"] == st_ref["rays"] and st["flagged"] == 0
    gs = {}
    counting = cam.render_rows(world, 0, shard, stats=gs)  # the reference-order counting kernel
    assert np.array_equal(ref, counting) and gs["rays"] == st_ref["rays"]
    ys = np.arange(0, cam.c.image_height, shard)[:: max(1, len(range(0, cam.c.image_height, shard)) // 6)].astype(np.uint32)
    gx, gy = np.meshgrid(np.arange(cam.c.image_width, dtype=np.uint32), ys)
    cpu = oracle.rtiow_render_pixels(world.desc, cam.c, gx.ravel(), gy.ravel()).reshape(len(ys), cam.c.image_width, 3)
    rows = (ys // shard).astype(int)
    assert np.abs(ref[rows] - cpu).max() <= 1e-9 * max(1.0, np.abs(cpu).max())
