"""Checkpoint / resume: the bincode layout of `Canvas` (ray-tracing-one-weekend/src/camera.rs:263-270,
examples/common/mod.rs:23-71) and Canvas::merge (camera.rs:273-291, test at :303-327)."""
import struct

import numpy as np
import pytest


def test_canvas_bincode_layout(rl):
    data = np.arange(2 * 3 * 3, dtype=np.float64).reshape(2, 3, 3) / 7.0
    c = rl.Canvas(10, 3, 2, data)
    b = c.to_bincode()
    assert len(b) == 32 + 6 * 24
    assert struct.unpack("<4Q", b[:32]) == (10, 3, 2, 6)  # samples, width, height, Vec len (u64 LE, fixed width)
    assert struct.unpack("<3d", b[32:56]) == tuple(data[0, 0])  # Color = [f64; 3], no per-element length
    back = rl.Canvas.from_bincode(b)
    assert (back.samples, back.width, back.height) == (10, 3, 2) and np.array_equal(back.data, data)
    with pytest.raises(ValueError):
        rl.Canvas.from_bincode(b[:-8])
    with pytest.raises(ValueError):
        rl.Canvas.from_bincode(b[:16])


def test_canvas_merge_known_answer(rl):
    # camera.rs:303-327 test_merge
    c1 = rl.Canvas(10, 100, 100, np.array([[1.0, 1.0, 1.0], [1.0, 1.0, 1.0]]))
    c2 = rl.Canvas(10, 100, 100, np.array([[2.0, 3.0, 4.0], [5.0, 6.0, 7.0]]))
    m = c1.merge(c2)
    assert m.samples == 20 and np.array_equal(m.data, np.array([[3.0, 4.0, 5.0], [6.0, 7.0, 8.0]]))


@pytest.mark.gpu
def test_resume_from_checkpoint_file_on_gpu(rl, oracle, tmp_path):
    world = rl.World.golden_test_scene()
    p = world.params
    p.samples_per_pixel, p.image_width = 5, 100
    cam = rl.Camera(p)
    first = cam.render(world)
    path = tmp_path / "render-0.chkpt"
    path.write_bytes(first.to_bincode())
    loaded = rl.Canvas.from_bincode(path.read_bytes())
    resumed = cam.render_from_checkpoint(world, loaded)  # samples 5..9 + merge
    in_memory = cam.render_from_checkpoint(world, first)
    assert resumed.samples == 10 and np.array_equal(resumed.data, in_memory.data)
    cpu_more = oracle.rtiow_render(world.desc, cam.c, first_sample=5)
    assert np.abs(resumed.data - (first.data + cpu_more)).max() <= 1e-9
