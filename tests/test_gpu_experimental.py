"""GPU tier, opt-in: the measured-and-lost kernel restructurings of csrc/experimental/ (wavefront form, pooled rays, two pixel
contexts per lane — DESIGN.md §3.5).  They are NOT in the product library; build `make -C rendering-learning_amd/csrc exp` and
run with RL_RENDER_LIB=.../librl_render_exp.so to check that they still render the product's bits."""
import numpy as np
import pytest

pytestmark = pytest.mark.gpu


def test_experimental_variants_equal_the_product_kernel(rl):
    if not rl.api.has_experimental():
        pytest.skip("product library loaded (experimental kernels live in librl_render_exp.so; set RL_RENDER_LIB)")
    world = rl.World.bouncing_spheres(1)
    p = world.params
    p.image_width, p.samples_per_pixel, p.max_depth = 120, 4, 50
    cam = rl.Camera(p)
    st0 = {}
    a = cam.render(world, stats=st0).data
    try:
        # wavefront, pool, wave2; the first correct kernel (v1: nested loops, exact divisions); the 3 / 2 / 1-waves layouts with the scene in LDS
        for v in (3, 5, 7, 1, 768, 512, 256):
            rl.api.set_rtiow_variant(v)
            sv = {}
            assert np.array_equal(cam.render(world, stats=sv).data, a), v
            for k in ("rays", "flagged") + (("node_tests", "sphere_tests", "rng_words") if v != 3 else ()):
                assert sv[k] == st0[k], (v, k)
    finally:
        rl.api.set_rtiow_variant(0)
