"""MI355X-native back end for the per-pixel / per-ray hot path of marcantony/rendering-learning.

Product = csrc/ (hand-written HIP for gfx950 behind the C ABI of include/rl_render.h) + host/ (C++
mirror of the reference's scene-building API).  `api` is the ctypes plumbing used by tests and bench.
"""
from . import api  # noqa: F401
from .api import (Camera, CameraParams, Canvas, RLError, RtcWorld, World, canvas_ppm, init, output_ppm,  # noqa: F401
                  rtc_camera)
