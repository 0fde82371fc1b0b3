"""ctypes bindings over the two in-tree native libraries of the product:

  csrc/librl_render.so  — the C-ABI drop-in boundary (include/rl_render.h): HIP kernels for gfx950.
  host/librl_host.so    — the C++ host mirror of the reference's scene-building API
                          (CameraParams/Camera::new, Sphere/Bvh/..., World, OBJ loaders, PPM writers).

Python here is plumbing only (tests, bench.py, torch.distributed): no arithmetic of the hot path
lives in this file, and there is NO CPU fallback — if librl_render.so is missing or no GPU is
present, every render call raises.

Class / function names mirror the reference:
  CameraParams, Camera(params).render(world) -> Canvas, render_from_checkpoint, Canvas.merge,
  output_ppm            <- ray-tracing-one-weekend/src/{camera.rs:23-143,263-296, output.rs:5}
  RtcCamera.render(world, aa) -> ppm via canvas_ppm
                        <- ray-tracer-challenge/src/scene/camera.rs:93, draw/canvas.rs:50
"""
import ctypes as C
import os
from dataclasses import dataclass, field

import numpy as np

_HERE = os.path.dirname(os.path.abspath(__file__))
# RL_RENDER_LIB selects another build of the same ABI (csrc/librl_render_exp.so: + the experimental kernel variants, A/B runs only)
RENDER_LIB = os.environ.get("RL_RENDER_LIB") or os.path.join(_HERE, "csrc", "librl_render.so")
HOST_LIB = os.path.join(_HERE, "host", "librl_host.so")

RL_OK, RL_E_INVALID, RL_E_NO_DEVICE, RL_E_DEVICE, RL_E_UNSUPPORTED, RL_E_DEGENERATE, RL_E_NOMEM = 0, -1, -2, -3, -4, -5, -6


class RLError(RuntimeError):
    def __init__(self, code, msg):
        super().__init__(f"rl error {code}: {msg}")
        self.code = code


# ----------------------------------------------------------------------------- C structs
class Stats(C.Structure):
    _fields_ = [("rays", C.c_uint64), ("node_tests", C.c_uint64), ("sphere_tests", C.c_uint64),
                ("planar_tests", C.c_uint64), ("instance_enters", C.c_uint64), ("rng_words", C.c_uint64),
                ("flagged", C.c_uint64), ("kernel_ms", C.c_double)]

    def as_dict(self):
        return {k: getattr(self, k) for k, _ in self._fields_}


class RtiowCamera(C.Structure):
    _fields_ = [("image_width", C.c_uint32), ("image_height", C.c_uint32),
                ("samples_per_pixel", C.c_uint32), ("max_depth", C.c_uint32),
                ("lookfrom", C.c_double * 3), ("pixel_00", C.c_double * 3),
                ("pixel_du", C.c_double * 3), ("pixel_dv", C.c_double * 3),
                ("defocus_disk_u", C.c_double * 3), ("defocus_disk_v", C.c_double * 3),
                ("defocus_angle", C.c_double), ("background", C.c_double * 3), ("seed", C.c_uint64)]


class RtcCamera(C.Structure):
    _fields_ = [("hsize", C.c_uint32), ("vsize", C.c_uint32), ("inverse", C.c_double * 16),
                ("pixel_size", C.c_double), ("half_width", C.c_double), ("half_height", C.c_double)]


class _HCameraParams(C.Structure):
    _fields_ = [("aspect_ratio", C.c_double), ("image_width", C.c_uint64), ("samples_per_pixel", C.c_uint64),
                ("max_depth", C.c_uint64), ("vfov", C.c_double), ("lookfrom", C.c_double * 3),
                ("lookat", C.c_double * 3), ("vup", C.c_double * 3), ("defocus_angle", C.c_double),
                ("focus_dist", C.c_double), ("background", C.c_double * 3), ("seed", C.c_uint64)]


# numpy dtypes of the POD scene records (include/rl_render.h) for Python-built scenes
HREF = np.dtype([("kind", "<u4"), ("index", "<u4")])
SPHERE = np.dtype([("center0", "<f8", 3), ("center1", "<f8", 3), ("radius", "<f8"), ("moving", "<u4"), ("material", "<u4")])
MATERIAL = np.dtype([("kind", "<u4"), ("texture", "<u4"), ("albedo", "<f8", 3), ("fuzz", "<f8"), ("ior", "<f8")])
TEXTURE = np.dtype([("kind", "<u4"), ("even", "<u4"), ("odd", "<u4"), ("image", "<u4"), ("color", "<f8", 3), ("inv_scale", "<f8")])
PERLIN = np.dtype([("randvec", "<f8", (256, 3)), ("perm_x", "<u4", 256), ("perm_y", "<u4", 256), ("perm_z", "<u4", 256)])
RTC_TRIANGLE = np.dtype([("p1", "<f8", 3), ("e1", "<f8", 3), ("e2", "<f8", 3), ("smooth", "<u4"), ("material", "<u4"),
                         ("n1", "<f8", 3), ("n2", "<f8", 3), ("n3", "<f8", 3)])
RTC_GROUP = np.dtype([("first", "<u4"), ("count", "<u4")])
RTC_BOUNDED = np.dtype([("minimum", "<f8", 3), ("maximum", "<f8", 3), ("child", HREF)])
RTC_TRANSFORMED = np.dtype([("inverse", "<f8", 16), ("inverse_transpose", "<f8", 16), ("child", HREF)])
RTC_MATERIAL = np.dtype([("color", "<f8", 3), ("ambient", "<f8"), ("diffuse", "<f8"), ("specular", "<f8"), ("shininess", "<f8"),
                         ("reflectivity", "<f8"), ("transparency", "<f8"), ("refractive_index", "<f8"), ("pattern", "<u4"), ("reserved", "<u4")])
RTC_SHAPE = np.dtype([("kind", "<u4"), ("material", "<u4"), ("has_minimum", "<u4"), ("has_maximum", "<u4"), ("closed", "<u4"), ("reserved", "<u4"),
                      ("minimum", "<f8"), ("maximum", "<f8")])
RTC_CSG = np.dtype([("operation", "<u4"), ("reserved", "<u4"), ("left", HREF), ("right", HREF)])
RTC_PATTERN = np.dtype([("kind", "<u4"), ("reserved", "<u4"), ("a", "<f8", 3), ("b", "<f8", 3), ("inverse", "<f8", 16)])
RTC_LIGHT = np.dtype([("position", "<f8", 3), ("intensity", "<f8", 3)])

MAT_FLAT, MAT_LAMBERTIAN, MAT_METAL, MAT_DIELECTRIC, MAT_DIFFUSE_LIGHT, MAT_ISOTROPIC = 0, 1, 2, 3, 4, 5
TEX_SOLID, TEX_CHECKER, TEX_IMAGE, TEX_NOISE = 0, 1, 2, 3
O_TRIANGLE, O_GROUP, O_BOUNDED, O_TRANSFORMED, O_SPHERE, O_PLANE, O_CUBE, O_CYLINDER, O_CONE, O_CSG = 1, 2, 3, 4, 5, 6, 7, 8, 9, 10
CSG_UNION, CSG_INTERSECTION, CSG_DIFFERENCE = 0, 1, 2
PAT_STRIPE, PAT_RING, PAT_GRADIENT, PAT_CHECKER3D = 1, 2, 3, 4


class RtcSceneDesc(C.Structure):
    _fields_ = [("triangles", C.c_void_p), ("n_triangles", C.c_uint32),
                ("groups", C.c_void_p), ("n_groups", C.c_uint32),
                ("group_items", C.c_void_p), ("n_group_items", C.c_uint32),
                ("boundeds", C.c_void_p), ("n_boundeds", C.c_uint32),
                ("transformeds", C.c_void_p), ("n_transformeds", C.c_uint32),
                ("materials", C.c_void_p), ("n_materials", C.c_uint32),
                ("objects", C.c_void_p), ("n_objects", C.c_uint32),
                ("lights", C.c_void_p), ("n_lights", C.c_uint32),
                ("max_reflection_depth", C.c_uint32), ("reserved", C.c_uint32),
                ("void_color", C.c_double * 3),
                ("shapes", C.c_void_p), ("n_shapes", C.c_uint32),
                ("csgs", C.c_void_p), ("n_csgs", C.c_uint32),
                ("patterns", C.c_void_p), ("n_patterns", C.c_uint32)]


# ----------------------------------------------------------------------------- library loading
_host = None
_render = None


def _one_hip_runtime():
    """PyTorch-ROCm wheels bundle their own libamdhip64.so.7; librl_render.so links the system one (/opt/rocm).  A process that
    ends up with BOTH sees the GPU only through whichever initialises first.  If torch is importable it is therefore imported
    BEFORE the product library is loaded: the dynamic loader then resolves librl_render.so's libamdhip64.so.7 to the copy torch
    already mapped (same SONAME) and the process has one runtime.  RL_NO_TORCH_PRELOAD=1 skips this (pure C / ctypes hosts)."""
    import sys
    if "torch" in sys.modules or os.environ.get("RL_NO_TORCH_PRELOAD"):
        return
    try:
        import torch  # noqa: F401
    except ImportError:
        pass


def host_lib():
    global _host
    if _host is None:
        _one_hip_runtime()  # librl_host.so links librl_render.so
        if not os.path.exists(HOST_LIB):
            raise RuntimeError(f"{HOST_LIB} not built — run `python -c 'import __graft_entry__ as g; g.build()'`")
        L = C.CDLL(HOST_LIB)
        L.rlh_last_error.restype = C.c_char_p
        for n in ("rlh_rtiow_golden_test_scene", "rlh_rtiow_bouncing_spheres", "rlh_rtiow_cow_scene", "rlh_rtiow_from_spheres",
                  "rlh_rtiow_desc", "rlh_rtc_test_obj_scene", "rlh_rtc_desc", "rlh_rtiow_output_ppm", "rlh_rtc_canvas_ppm"):
            getattr(L, n).restype = C.c_void_p
        L.rlh_rtiow_bouncing_spheres.argtypes = [C.c_uint64]
        L.rlh_rtiow_cow_scene.argtypes = [C.c_char_p, C.c_uint64, C.c_void_p, C.c_uint32, C.c_uint32]
        L.rlh_rtiow_from_spheres.argtypes = [C.c_void_p, C.c_uint32, C.c_void_p, C.c_uint32, C.c_void_p, C.c_uint32, C.c_int]
        L.rlh_rtiow_desc.argtypes = [C.c_void_p]
        L.rlh_rtiow_free.argtypes = [C.c_void_p]
        L.rlh_rtiow_get_params.argtypes = [C.c_void_p, C.POINTER(_HCameraParams)]
        L.rlh_rtiow_camera_new.argtypes = [C.POINTER(_HCameraParams), C.POINTER(RtiowCamera)]
        L.rlh_rtiow_output_ppm.argtypes = [C.c_void_p, C.c_uint64, C.c_uint64, C.c_uint64, C.POINTER(C.c_uint64)]
        L.rlh_free.argtypes = [C.c_void_p]
        L.rlh_rtc_test_obj_scene.argtypes = [C.c_char_p, C.c_uint64, C.c_uint64, C.c_uint64]
        L.rlh_rtc_desc.argtypes = [C.c_void_p]
        L.rlh_rtc_get_camera.argtypes = [C.c_void_p, C.POINTER(RtcCamera)]
        L.rlh_rtc_free.argtypes = [C.c_void_p]
        L.rlh_rtc_camera_new.argtypes = [C.c_uint64, C.c_uint64, C.c_double, C.c_void_p, C.c_void_p, C.c_void_p, C.POINTER(RtcCamera)]
        L.rlh_rtc_camera_from_matrix.argtypes = [C.c_uint64, C.c_uint64, C.c_double, C.c_void_p, C.POINTER(RtcCamera)]
        L.rlh_rtc_make_transformed.argtypes = [C.c_void_p, C.c_void_p]
        L.rlh_rtc_rotation.argtypes = [C.c_int, C.c_double, C.c_void_p]
        L.rlh_rtc_matmul.argtypes = [C.c_void_p, C.c_void_p, C.c_void_p]
        L.rlh_rtc_invert.argtypes = [C.c_void_p, C.c_void_p]
        L.rlh_rtc_canvas_ppm.argtypes = [C.c_void_p, C.c_uint64, C.c_uint64, C.POINTER(C.c_uint64)]
        _host = L
    return _host


# every symbol include/rl_render.h declares (checked by tests/test_abi.py)
RENDER_SYMBOLS = ["rl_init", "rl_init_multi", "rl_device_count", "rl_shutdown", "rl_last_error", "rl_abi_version", "rl_device_info", "rl_rtiow_render_progress",
                  "rl_scene_destroy", "rl_render_status",
                  "rl_rtiow_scene_create", "rl_bvh_build", "rl_rtiow_render", "rl_rtiow_render_rows", "rl_rtiow_render_device",
                  "rl_rtiow_render_multi", "rl_rtiow_render_multi_device", "rl_rtiow_encode_rgb8_device", "rl_rtiow_render_rgb8",
                  "rl_rtc_scene_create", "rl_rtc_render", "rl_rtc_render_rows", "rl_rtc_render_device",
                  "rl_rtc_render_multi", "rl_rtc_render_multi_device", "rl_rtc_encode_rgb8_device", "rl_rtc_render_rgb8"]


def render_lib():
    """The HIP product library. Fails loudly when it is not built."""
    global _render
    if _render is None:
        _one_hip_runtime()
        if not os.path.exists(RENDER_LIB):
            raise RuntimeError(f"{RENDER_LIB} not built — the HIP extension is required (no CPU fallback); "
                               "run `python -c 'import __graft_entry__ as g; g.build()'`")
        L = C.CDLL(RENDER_LIB)
        L.rl_last_error.restype = C.c_char_p
        L.rl_init.argtypes = [C.c_int]
        L.rl_device_info.argtypes = [C.c_char_p, C.c_int]
        L.rl_scene_destroy.argtypes = [C.c_void_p]
        L.rl_rtiow_scene_create.restype = C.c_void_p
        L.rl_rtiow_scene_create.argtypes = [C.c_void_p]
        L.rl_rtc_scene_create.restype = C.c_void_p
        L.rl_rtc_scene_create.argtypes = [C.c_void_p]
        L.rl_rtiow_render.argtypes = [C.c_void_p, C.POINTER(RtiowCamera), C.c_uint64, C.c_void_p, C.POINTER(Stats)]
        L.rl_rtiow_render_rows.argtypes = [C.c_void_p, C.POINTER(RtiowCamera), C.c_uint64, C.c_uint32, C.c_uint32, C.c_void_p, C.POINTER(Stats)]
        L.rl_rtiow_render_device.argtypes = [C.c_void_p, C.POINTER(RtiowCamera), C.c_uint64, C.c_uint32, C.c_uint32, C.c_void_p, C.c_void_p, C.POINTER(Stats)]
        L.rl_rtiow_render_rgb8.argtypes = [C.c_void_p, C.POINTER(RtiowCamera), C.c_uint64, C.c_void_p, C.POINTER(Stats)]
        L.rl_rtiow_encode_rgb8_device.argtypes = [C.c_void_p, C.c_uint64, C.c_uint32, C.c_void_p, C.c_void_p]
        L.rl_rtc_render_rgb8.argtypes = [C.c_void_p, C.POINTER(RtcCamera), C.c_uint32, C.c_void_p, C.POINTER(Stats)]
        L.rl_rtc_encode_rgb8_device.argtypes = [C.c_void_p, C.c_uint64, C.c_void_p, C.c_void_p]
        L.rl_rtc_render.argtypes = [C.c_void_p, C.POINTER(RtcCamera), C.c_uint32, C.c_void_p, C.POINTER(Stats)]
        L.rl_rtc_render_rows.argtypes = [C.c_void_p, C.POINTER(RtcCamera), C.c_uint32, C.c_uint32, C.c_uint32, C.c_void_p, C.POINTER(Stats)]
        L.rl_rtc_render_device.argtypes = [C.c_void_p, C.POINTER(RtcCamera), C.c_uint32, C.c_uint32, C.c_uint32, C.c_void_p, C.c_void_p, C.POINTER(Stats)]
        L.rl_init_multi.argtypes = [C.c_int]
        L.rl_render_status.argtypes = [C.c_void_p, C.POINTER(Stats)]
        L.rl_rtiow_render_multi.argtypes = [C.c_void_p, C.POINTER(RtiowCamera), C.c_uint64, C.c_void_p, C.POINTER(Stats)]
        L.rl_rtiow_render_multi_device.argtypes = [C.c_void_p, C.POINTER(RtiowCamera), C.c_uint64, C.c_void_p, C.POINTER(Stats)]
        L.rl_rtc_render_multi.argtypes = [C.c_void_p, C.POINTER(RtcCamera), C.c_uint32, C.c_void_p, C.POINTER(Stats)]
        L.rl_rtc_render_multi_device.argtypes = [C.c_void_p, C.POINTER(RtcCamera), C.c_uint32, C.c_void_p, C.POINTER(Stats)]
        _render = L
    return _render


_inited = False


def init(device=-1):
    global _inited
    L = render_lib()
    rc = L.rl_init(int(device))
    if rc != RL_OK:
        raise RLError(rc, L.rl_last_error().decode())
    _inited = True


def init_multi(n_devices=0, emulate=0):
    """rl_init_multi: one process drives n_devices GPUs (0 = all).  emulate=G (tests on a one-GPU box): G device contexts on the
    current GPU.  Scenes must be (re)created afterwards so that every context holds a replica."""
    global _inited
    L = render_lib()
    if emulate:
        L.rl_debug_init_multi_emulated.argtypes = [C.c_int]
        rc = L.rl_debug_init_multi_emulated(int(emulate))
    else:
        rc = L.rl_init_multi(int(n_devices))
    if rc != RL_OK:
        raise RLError(rc, L.rl_last_error().decode())
    _inited = True
    return L.rl_device_count()


def render_status(world, allow_degenerate=False):
    """rl_render_status: waits for the scene's last asynchronous render; {'rays', 'flagged', 'rc'}."""
    st = Stats()
    rc = render_lib().rl_render_status(world.device(), C.byref(st))
    _check(rc, allow_degenerate)
    L = render_lib()
    L.rl_debug_slow_traces.restype = C.c_uint64
    return {"rays": st.rays, "flagged": st.flagged, "rc": rc, "slow_traces": L.rl_debug_slow_traces()}


def render_progress(world):
    """rl_rtiow_render_progress: (pixel slots claimed so far in the running launch, slots of that launch, phase) — does not wait for the render."""
    L = render_lib()
    a, b, ph = C.c_uint64(), C.c_uint64(), C.c_uint32()
    L.rl_rtiow_render_progress.argtypes = [C.c_void_p, C.POINTER(C.c_uint64), C.POINTER(C.c_uint64), C.POINTER(C.c_uint32)]
    _check(L.rl_rtiow_render_progress(world.device(), C.byref(a), C.byref(b), C.byref(ph)))
    return a.value, b.value, ph.value


def set_fast_traversal(on):
    """Tests / tools: counter-free renders of small sphere scenes use the fast (ordered, reject-only) traversal unless switched off."""
    render_lib().rl_debug_set_fast_traversal(int(bool(on)))


def set_coop(on):
    """Tests / tools: counter-free renders of SMALL frames of sphere scenes use the cooperative one-wave-per-pixel kernel unless switched off."""
    render_lib().rl_debug_set_coop(int(bool(on)))


def set_steal(max_fill):
    """Tests / tools: work stealing on small shards (at most max_fill x as many pixels as the GPU has lanes; 0 switches it off)."""
    L = render_lib()
    L.rl_debug_set_steal.argtypes = [C.c_double]
    L.rl_debug_set_steal(float(max_fill))


def has_experimental():
    return bool(render_lib().rl_debug_has_experimental())


def _check(rc, allow_degenerate=False):
    if rc == RL_OK or (allow_degenerate and rc == RL_E_DEGENERATE):
        return rc
    raise RLError(rc, render_lib().rl_last_error().decode())


def rows_for(height, row_first, row_step):
    return 0 if row_first >= height else (height - row_first + row_step - 1) // row_step


# ----------------------------------------------------------------------------- RTIOW host mirror
@dataclass
class CameraParams:  # camera.rs:23-59, defaults as in the reference
    aspect_ratio: float = 1.0
    image_width: int = 100
    samples_per_pixel: int = 10
    max_depth: int = 10
    vfov: float = 90.0
    lookfrom: tuple = (0.0, 0.0, 0.0)
    lookat: tuple = (0.0, 0.0, -1.0)
    vup: tuple = (0.0, 1.0, 0.0)
    defocus_angle: float = 0.0
    focus_dist: float = 10.0
    background: tuple = (0.7, 0.8, 1.0)
    seed: int = 0

    def _c(self):
        p = _HCameraParams()
        p.aspect_ratio, p.image_width, p.samples_per_pixel, p.max_depth = self.aspect_ratio, self.image_width, self.samples_per_pixel, self.max_depth
        p.vfov, p.defocus_angle, p.focus_dist, p.seed = self.vfov, self.defocus_angle, self.focus_dist, self.seed
        p.lookfrom[:], p.lookat[:], p.vup[:], p.background[:] = self.lookfrom, self.lookat, self.vup, self.background
        return p

    @staticmethod
    def _from_c(p):
        return CameraParams(p.aspect_ratio, p.image_width, p.samples_per_pixel, p.max_depth, p.vfov, tuple(p.lookfrom),
                            tuple(p.lookat), tuple(p.vup), p.defocus_angle, p.focus_dist, tuple(p.background), p.seed)


class World:
    """A flattened RTIOW world (host arrays owned by librl_host) — what `world: H` is in camera.rs:122."""

    def __init__(self, handle):
        if not handle:
            raise RuntimeError("host scene build failed: " + host_lib().rlh_last_error().decode())
        self._h = handle
        self.desc = host_lib().rlh_rtiow_desc(handle)
        p = _HCameraParams()
        host_lib().rlh_rtiow_get_params(handle, C.byref(p))
        self.params = CameraParams._from_c(p)  # the example's / test's camera parameters
        self._device = None

    def counts(self):
        """Element counts of the flattened scene (rl_rtiow_scene_desc)."""
        out = (C.c_uint32 * 9)()
        L = host_lib()
        L.rlh_rtiow_counts.argtypes = [C.c_void_p, C.c_void_p]
        L.rlh_rtiow_counts(self._h, out)
        return dict(zip(("spheres", "planars", "media", "translates", "transforms", "lists", "bvh_nodes", "materials", "textures"), list(out)))

    def __del__(self):
        try:
            if self._device is not None:
                render_lib().rl_scene_destroy(self._device)
            host_lib().rlh_rtiow_free(self._h)
        except Exception:
            pass

    @staticmethod
    def golden_test_scene():  # tests/ray_tracing_one_weekend.rs:14-75
        return World(host_lib().rlh_rtiow_golden_test_scene())

    @staticmethod
    def bouncing_spheres(master_seed=1):  # examples/bouncing_spheres.rs
        return World(host_lib().rlh_rtiow_bouncing_spheres(master_seed))

    @staticmethod
    def cow_scene(obj_text: bytes, rgb8: np.ndarray):  # examples/cow.rs
        rgb8 = np.ascontiguousarray(rgb8, dtype=np.uint8)
        h, w = rgb8.shape[:2]
        return World(host_lib().rlh_rtiow_cow_scene(obj_text, len(obj_text), rgb8.ctypes.data, w, h))

    @staticmethod
    def perlin_spheres():  # examples/perlin_spheres.rs
        L = host_lib()
        L.rlh_rtiow_perlin_scene.restype, L.rlh_rtiow_perlin_scene.argtypes = C.c_void_p, [C.c_int]
        return World(L.rlh_rtiow_perlin_scene(0))

    @staticmethod
    def simple_light():  # examples/simple_light.rs
        L = host_lib()
        L.rlh_rtiow_perlin_scene.restype, L.rlh_rtiow_perlin_scene.argtypes = C.c_void_p, [C.c_int]
        return World(L.rlh_rtiow_perlin_scene(1))

    @staticmethod
    def earth_scene(rgb8: np.ndarray):  # examples/earth.rs with the caller's image (sRGB8, [H, W, 3])
        L = host_lib()
        L.rlh_rtiow_earth_scene.restype, L.rlh_rtiow_earth_scene.argtypes = C.c_void_p, [C.c_void_p, C.c_uint32, C.c_uint32]
        rgb8 = np.ascontiguousarray(rgb8, dtype=np.uint8)
        h, w = rgb8.shape[:2]
        return World(L.rlh_rtiow_earth_scene(rgb8.ctypes.data, w, h))

    @staticmethod
    def example_scene(name: str, obj_text: bytes = None, rgb8: np.ndarray = None):
        """The reference's other example scenes (host/scenes.hpp): "checkered_spheres", "quads", "flat_world", "cornell_box",
        "cornell_smoke", "teapot" (obj_text = teapot-low.obj), "final_scene" (rgb8 = the earth image, sRGB8 [H, W, 3])."""
        L = host_lib()
        L.rlh_rtiow_example_scene.restype = C.c_void_p
        L.rlh_rtiow_example_scene.argtypes = [C.c_char_p, C.c_char_p, C.c_uint64, C.c_void_p, C.c_uint32, C.c_uint32]
        h = w = 0
        ptr = None
        if rgb8 is not None:
            rgb8 = np.ascontiguousarray(rgb8, dtype=np.uint8)
            h, w = rgb8.shape[:2]
            ptr = rgb8.ctypes.data
        return World(L.rlh_rtiow_example_scene(name.encode(), obj_text, len(obj_text) if obj_text else 0, ptr, w, h))

    @staticmethod
    def stress_scene(n_side=1000, subdiv=2, obj_text: bytes = None, rgb8: np.ndarray = None, seed=5, device_bvh=False):
        """BASELINE configs[4]: n_side^2 small spheres + ground + subdivided spot mesh (see host/scenes.hpp).
        device_bvh: build both BVHs with rl_bvh_build on the GPU instead of the host recursion (same tree)."""
        L = host_lib()
        L.rlh_rtiow_stress_scene.restype = C.c_void_p
        L.rlh_rtiow_stress_scene.argtypes = [C.c_int, C.c_int, C.c_char_p, C.c_uint64, C.c_void_p, C.c_uint32, C.c_uint32, C.c_uint64, C.c_int]
        if device_bvh and not _inited:
            init()
        if obj_text is None:
            return World(L.rlh_rtiow_stress_scene(n_side, subdiv, None, 0, None, 0, 0, seed, int(device_bvh)))
        rgb8 = np.ascontiguousarray(rgb8, dtype=np.uint8)
        h, w = rgb8.shape[:2]
        return World(L.rlh_rtiow_stress_scene(n_side, subdiv, obj_text, len(obj_text), rgb8.ctypes.data, w, h, seed, int(device_bvh)))

    @staticmethod
    def from_spheres(spheres, materials, textures, use_bvh):
        spheres = np.ascontiguousarray(spheres, dtype=SPHERE)
        materials = np.ascontiguousarray(materials, dtype=MATERIAL)
        textures = np.ascontiguousarray(textures, dtype=TEXTURE)
        return World(host_lib().rlh_rtiow_from_spheres(spheres.ctypes.data, len(spheres), materials.ctypes.data, len(materials),
                                                       textures.ctypes.data, len(textures), 1 if use_bvh else 0))

    @staticmethod
    def build(fn):
        """Compose a world with the reference's scene-building vocabulary: fn(SceneBuilder) -> root object id."""
        b = SceneBuilder()
        try:
            root = fn(b)
            return World(host_lib().rlh_b_finish(b._b, root))
        finally:
            host_lib().rlh_builder_free(b._b)

    def device(self):
        """rl_rtiow_scene_create — uploads once, cached."""
        if self._device is None:
            if not _inited:
                init()
            L = render_lib()
            h = L.rl_rtiow_scene_create(self.desc)
            if not h:
                raise RLError(RL_E_INVALID, L.rl_last_error().decode())
            self._device = h
        return self._device


class SceneBuilder:
    """Thin handle over librl_host's builder: textures, materials and hittables by id (names follow the reference)."""

    def __init__(self):
        L = host_lib()
        L.rlh_builder_new.restype = C.c_void_p
        for n, a in (("rlh_builder_free", [C.c_void_p]), ("rlh_b_solid", [C.c_void_p, C.c_void_p]),
                     ("rlh_b_checker", [C.c_void_p, C.c_double, C.c_int, C.c_int]),
                     ("rlh_b_image", [C.c_void_p, C.c_void_p, C.c_uint32, C.c_uint32]),
                     ("rlh_b_noise", [C.c_void_p, C.c_double, C.c_uint64]),
                     ("rlh_b_material", [C.c_void_p, C.c_uint32, C.c_int, C.c_void_p, C.c_double, C.c_double]),
                     ("rlh_b_sphere", [C.c_void_p, C.c_void_p, C.c_void_p, C.c_double, C.c_int]),
                     ("rlh_b_planar", [C.c_void_p, C.c_uint32, C.c_void_p, C.c_void_p, C.c_void_p, C.c_int]),
                     ("rlh_b_triangle", [C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p, C.c_int]),
                     ("rlh_b_translate", [C.c_void_p, C.c_int, C.c_void_p]),
                     ("rlh_b_medium", [C.c_void_p, C.c_int, C.c_double, C.c_int]),
                     ("rlh_b_transform", [C.c_void_p, C.c_int, C.c_int, C.c_double]),
                     ("rlh_b_group", [C.c_void_p, C.c_void_p, C.c_uint32, C.c_int]),
                     ("rlh_b_obj", [C.c_void_p, C.c_char_p, C.c_uint64, C.c_int]),
                     ("rlh_b_finish", [C.c_void_p, C.c_int])):
            getattr(L, n).argtypes = a
        L.rlh_b_finish.restype = C.c_void_p
        self._L = L
        self._b = L.rlh_builder_new()

    @staticmethod
    def _v(x):
        return None if x is None else np.ascontiguousarray(x, dtype=np.float64)

    def _chk(self, r):
        if r < 0:
            raise RuntimeError("scene builder: " + self._L.rlh_last_error().decode())
        return r

    def solid(self, color):
        c = self._v(color)
        return self._chk(self._L.rlh_b_solid(self._b, c.ctypes.data))

    def checker(self, scale, even, odd):
        return self._chk(self._L.rlh_b_checker(self._b, scale, even, odd))

    def image(self, rgb_f32):
        a = np.ascontiguousarray(rgb_f32, dtype=np.float32)
        return self._chk(self._L.rlh_b_image(self._b, a.ctypes.data, a.shape[1], a.shape[0]))

    def noise(self, scale, seed):
        """Noise{Perlin::new(&mut Xoshiro256PlusPlus::seed_from_u64(seed)), scale} (texture.rs:84)."""
        return self._chk(self._L.rlh_b_noise(self._b, scale, seed))

    def lambertian(self, tex):
        return self._chk(self._L.rlh_b_material(self._b, MAT_LAMBERTIAN, tex, None, 0.0, 1.0))

    def metal(self, albedo, fuzz):
        a = self._v(albedo)
        return self._chk(self._L.rlh_b_material(self._b, MAT_METAL, -1, a.ctypes.data, fuzz, 1.0))

    def dielectric(self, ior):
        return self._chk(self._L.rlh_b_material(self._b, MAT_DIELECTRIC, -1, None, 0.0, ior))

    def diffuse_light(self, tex):
        return self._chk(self._L.rlh_b_material(self._b, MAT_DIFFUSE_LIGHT, tex, None, 0.0, 1.0))

    def flat(self):
        return self._chk(self._L.rlh_b_material(self._b, MAT_FLAT, -1, None, 0.0, 1.0))

    def isotropic(self, tex):
        return self._chk(self._L.rlh_b_material(self._b, MAT_ISOTROPIC, tex, None, 0.0, 1.0))

    def constant_medium(self, boundary, density, mat):
        """ConstantMedium::new(boundary, density, phase_function) — deterministic variant (rl_render.h rl_medium)."""
        return self._chk(self._L.rlh_b_medium(self._b, boundary, density, mat))

    def sphere(self, center, radius, mat, center2=None):
        c0, c1 = self._v(center), self._v(center2)
        return self._chk(self._L.rlh_b_sphere(self._b, c0.ctypes.data, None if c1 is None else c1.ctypes.data, radius, mat))

    def _planar(self, kind, q, u, v, mat):
        q, u, v = self._v(q), self._v(u), self._v(v)
        return self._chk(self._L.rlh_b_planar(self._b, kind, q.ctypes.data, u.ctypes.data, v.ctypes.data, mat))

    def plane(self, q, u, v, mat):
        return self._planar(0, q, u, v, mat)

    def quad(self, q, u, v, mat):
        return self._planar(1, q, u, v, mat)

    def triangle(self, q, u, v, mat):
        return self._planar(2, q, u, v, mat)

    def triangle_from_model(self, points, mat, uvs=None, normals=None):
        p, t, n = self._v(points), self._v(uvs), self._v(normals)
        return self._chk(self._L.rlh_b_triangle(self._b, p.ctypes.data, None if t is None else t.ctypes.data,
                                                None if n is None else n.ctypes.data, mat))

    def translate(self, obj, offset):
        o = self._v(offset)
        return self._chk(self._L.rlh_b_translate(self._b, obj, o.ctypes.data))

    def rotate_x(self, obj, deg):
        return self._chk(self._L.rlh_b_transform(self._b, obj, 0, deg))

    def rotate_y(self, obj, deg):
        return self._chk(self._L.rlh_b_transform(self._b, obj, 1, deg))

    def rotate_z(self, obj, deg):
        return self._chk(self._L.rlh_b_transform(self._b, obj, 2, deg))

    def scale(self, obj, s):
        return self._chk(self._L.rlh_b_transform(self._b, obj, 3, s))

    def bvh(self, objs):
        a = np.ascontiguousarray(objs, dtype=np.int32)
        return self._chk(self._L.rlh_b_group(self._b, a.ctypes.data, len(a), 1))

    def list(self, objs):
        a = np.ascontiguousarray(objs, dtype=np.int32)
        return self._chk(self._L.rlh_b_group(self._b, a.ctypes.data, len(a), 0))

    def obj_mesh(self, obj_text: bytes, mat):
        return self._chk(self._L.rlh_b_obj(self._b, obj_text, len(obj_text), mat))


def set_rtiow_variant(v):
    """Tests / tools: force a kernel variant (0 auto, 1 nested-loop, 2 general, 512/768/1024 wave)."""
    L = render_lib()
    L.rl_debug_set_rtiow_variant.argtypes = [C.c_int]
    L.rl_debug_set_rtiow_variant(int(v))


@dataclass
class Canvas:  # camera.rs:263-296; data = SUMS over samples, [H, W, 3] f64
    samples: int
    width: int
    height: int
    data: np.ndarray = field(repr=False)

    def merge(self, other):  # camera.rs:273-291
        assert self.width == other.width and self.height == other.height and self.data.shape == other.data.shape
        return Canvas(self.samples + other.samples, self.width, self.height, self.data + other.data)

    def pixel_data(self):  # camera.rs:293: c / samples == c * (1/samples)
        return self.data * (1.0 / self.samples)

    def to_bincode(self) -> bytes:
        """The reference's checkpoint bytes: bincode::serialize(&canvas) (examples/common/mod.rs:32)."""
        L = host_lib()
        L.rlh_canvas_to_bincode.restype = C.c_void_p
        L.rlh_canvas_to_bincode.argtypes = [C.c_uint64, C.c_uint64, C.c_uint64, C.c_void_p, C.c_uint64, C.POINTER(C.c_uint64)]
        d = np.ascontiguousarray(self.data, dtype=np.float64)
        n = C.c_uint64()
        p = L.rlh_canvas_to_bincode(self.samples, self.width, self.height, d.ctypes.data, d.size // 3, C.byref(n))
        b = C.string_at(p, n.value)
        L.rlh_free(p)
        return b

    @staticmethod
    def from_bincode(b: bytes):
        """bincode::deserialize::<Canvas> (examples/common/mod.rs:45-46)."""
        L = host_lib()
        L.rlh_canvas_from_bincode.argtypes = [C.c_char_p, C.c_uint64] + [C.POINTER(C.c_uint64)] * 4
        s, w, h, n = C.c_uint64(), C.c_uint64(), C.c_uint64(), C.c_uint64()
        if L.rlh_canvas_from_bincode(b, len(b), C.byref(s), C.byref(w), C.byref(h), C.byref(n)) != 0:
            raise ValueError("malformed checkpoint: " + L.rlh_last_error().decode())
        data = np.frombuffer(b, dtype="<f8", offset=32, count=n.value * 3).copy()
        if n.value == w.value * h.value:
            data = data.reshape(h.value, w.value, 3)
        else:
            data = data.reshape(-1, 3)
        return Canvas(s.value, w.value, h.value, data)


class Camera:
    def __init__(self, params: CameraParams):  # Camera::new camera.rs:72
        self.params = params
        self.c = RtiowCamera()
        pc = params._c()
        if host_lib().rlh_rtiow_camera_new(C.byref(pc), C.byref(self.c)) != 0:
            raise RuntimeError("Camera::new: " + host_lib().rlh_last_error().decode())
        self.image_height = self.c.image_height

    def _render(self, first_sample, world: World, row_first=0, row_step=1, stats=None, allow_degenerate=False):
        nrows = rows_for(self.c.image_height, row_first, row_step)
        out = np.empty((nrows, self.c.image_width, 3), dtype=np.float64)
        st = Stats()
        rc = render_lib().rl_rtiow_render_rows(world.device(), C.byref(self.c), first_sample, row_first, row_step, out.ctypes.data, C.byref(st))
        _check(rc, allow_degenerate)
        if stats is not None:
            stats.update(st.as_dict())
            stats["rc"] = rc
        return out

    def render(self, world: World, stats=None, allow_degenerate=False) -> Canvas:  # camera.rs:122
        data = self._render(0, world, stats=stats, allow_degenerate=allow_degenerate)
        return Canvas(self.params.samples_per_pixel, self.c.image_width, self.c.image_height, data)

    def render_rgb8(self, world: World) -> np.ndarray:
        """render + the device output stage (sRGB, floor(v*255.999)): the [H, W, 3] bytes output_ppm prints."""
        out = np.empty((self.c.image_height, self.c.image_width, 3), dtype=np.uint8)
        _check(render_lib().rl_rtiow_render_rgb8(world.device(), C.byref(self.c), 0, out.ctypes.data, None), allow_degenerate=True)
        return out

    def render_from_checkpoint(self, world: World, checkpoint: Canvas) -> Canvas:  # camera.rs:136-143
        data = self._render(checkpoint.samples, world)
        return Canvas(self.params.samples_per_pixel, self.c.image_width, self.c.image_height, data).merge(checkpoint)

    def render_rows(self, world: World, row_first, row_step, first_sample=0, stats=None):
        return self._render(first_sample, world, row_first, row_step, stats)

    def render_multi(self, world: World, first_sample=0, stats=None, allow_degenerate=False) -> Canvas:
        """rl_rtiow_render_multi: the whole frame over every GPU of init_multi (rows interleaved, one RCCL exchange)."""
        out = np.empty((self.c.image_height, self.c.image_width, 3), dtype=np.float64)
        st = Stats()
        rc = render_lib().rl_rtiow_render_multi(world.device(), C.byref(self.c), first_sample, out.ctypes.data, C.byref(st))
        _check(rc, allow_degenerate)
        if stats is not None:
            stats.update(st.as_dict())
            stats["rc"] = rc
        return Canvas(self.params.samples_per_pixel, self.c.image_width, self.c.image_height, out)

    def render_multi_device(self, world: World, d_ptr, first_sample=0, stats=None):
        """Frame left in GPU 0's HBM; asynchronous unless stats is a dict (api.render_status(world) waits)."""
        st = Stats() if stats is not None else None
        rc = render_lib().rl_rtiow_render_multi_device(world.device(), C.byref(self.c), first_sample, C.c_void_p(d_ptr), C.byref(st) if st is not None else None)
        _check(rc)
        if stats is not None:
            stats.update(st.as_dict())

    def render_device(self, world: World, d_ptr, stream=0, row_first=0, row_step=1, first_sample=0, stats=None):
        """Output stays in HBM: d_ptr = device pointer of nrows*W*3 f64. Async unless stats is a dict."""
        st = Stats() if stats is not None else None
        rc = render_lib().rl_rtiow_render_device(world.device(), C.byref(self.c), first_sample, row_first, row_step,
                                                 C.c_void_p(d_ptr), C.c_void_p(stream), C.byref(st) if st is not None else None)
        _check(rc)
        if stats is not None:
            stats.update(st.as_dict())


def _take_string(ptr, n):
    s = C.string_at(ptr, n.value)
    host_lib().rlh_free(ptr)
    return s.decode("ascii")


def output_ppm(canvas_or_sums, samples=None) -> str:  # output.rs:5-14
    if isinstance(canvas_or_sums, Canvas):
        data, samples = canvas_or_sums.data, canvas_or_sums.samples
    else:
        data = canvas_or_sums
    data = np.ascontiguousarray(data, dtype=np.float64)
    h, w = data.shape[:2]
    n = C.c_uint64()
    return _take_string(host_lib().rlh_rtiow_output_ppm(data.ctypes.data, w, h, samples, C.byref(n)), n)


# ----------------------------------------------------------------------------- RTC host mirror
class RtcWorld:
    """A flattened ray-tracer-challenge World (scene/world.rs:26) + the scene's Camera."""

    def __init__(self, handle=None, desc_struct=None, keep=None, camera=None):
        self._h = handle
        self._keep = keep
        self._device = None
        if handle is not None:
            if not handle:
                raise RuntimeError("host scene build failed: " + host_lib().rlh_last_error().decode())
            self.desc = host_lib().rlh_rtc_desc(handle)
            self.camera = RtcCamera()
            host_lib().rlh_rtc_get_camera(handle, C.byref(self.camera))
        else:
            self._desc_struct = desc_struct
            self.desc = C.addressof(desc_struct)
            self.camera = camera

    def __del__(self):
        try:
            if self._device is not None:
                render_lib().rl_scene_destroy(self._device)
            if self._h:
                host_lib().rlh_rtc_free(self._h)
        except Exception:
            pass

    @staticmethod
    def test_obj_scene(obj_text: bytes, res_x=300, res_y=200):  # tests/ray_tracer.rs:242-275
        return RtcWorld(host_lib().rlh_rtc_test_obj_scene(obj_text, len(obj_text), res_x, res_y))

    @staticmethod
    def test_mirror_scene(res_x=300, res_y=200):  # tests/ray_tracer.rs:56-240
        L = host_lib()
        L.rlh_rtc_named_scene.restype = C.c_void_p
        L.rlh_rtc_named_scene.argtypes = [C.c_int, C.c_uint64, C.c_uint64]
        return RtcWorld(L.rlh_rtc_named_scene(0, res_x, res_y))

    @staticmethod
    def test_csg_scene(res_x=300, res_y=200):  # tests/ray_tracer.rs:277-368
        L = host_lib()
        L.rlh_rtc_named_scene.restype = C.c_void_p
        L.rlh_rtc_named_scene.argtypes = [C.c_int, C.c_uint64, C.c_uint64]
        return RtcWorld(L.rlh_rtc_named_scene(1, res_x, res_y))

    @staticmethod
    def from_arrays(triangles, materials, objects, lights, groups=(), group_items=(), boundeds=(), transformeds=(),
                    max_reflection_depth=5, void_color=(0.0, 0.0, 0.0), camera=None, shapes=(), csgs=(), patterns=()):
        def arr(x, dt):
            return np.zeros(0, dtype=dt) if len(x) == 0 else np.ascontiguousarray(x, dtype=dt)
        arrs = dict(triangles=arr(triangles, RTC_TRIANGLE), groups=arr(groups, RTC_GROUP), group_items=arr(group_items, HREF),
                    boundeds=arr(boundeds, RTC_BOUNDED), transformeds=arr(transformeds, RTC_TRANSFORMED),
                    materials=arr(materials, RTC_MATERIAL), objects=arr(objects, HREF), lights=arr(lights, RTC_LIGHT),
                    shapes=arr(shapes, RTC_SHAPE), csgs=arr(csgs, RTC_CSG), patterns=arr(patterns, RTC_PATTERN))
        d = RtcSceneDesc()
        for k, a in arrs.items():
            setattr(d, k, a.ctypes.data if len(a) else None)
            setattr(d, "n_" + k, len(a))
        d.max_reflection_depth = max_reflection_depth
        d.void_color[:] = void_color
        return RtcWorld(desc_struct=d, keep=arrs, camera=camera)

    def device(self):
        if self._device is None:
            if not _inited:
                init()
            L = render_lib()
            h = L.rl_rtc_scene_create(self.desc)
            if not h:
                raise RLError(RL_E_INVALID, L.rl_last_error().decode())
            self._device = h
        return self._device

    def render(self, aa_samples=1, camera=None, row_first=0, row_step=1, stats=None, allow_degenerate=False):
        """Camera::render(&world, &RenderOpts{anti_aliasing_samples}) on the GPU -> [rows, W, 3] means."""
        cam = camera or self.camera
        nrows = rows_for(cam.vsize, row_first, row_step)
        out = np.empty((nrows, cam.hsize, 3), dtype=np.float64)
        st = Stats()
        rc = render_lib().rl_rtc_render_rows(self.device(), C.byref(cam), aa_samples, row_first, row_step, out.ctypes.data, C.byref(st))
        _check(rc, allow_degenerate)
        if stats is not None:
            stats.update(st.as_dict())
            stats["rc"] = rc
        return out

    def render_rgb8(self, aa_samples=1, camera=None) -> np.ndarray:
        """render + the device output stage (round(c*255)): the [H, W, 3] bytes Canvas::ppm prints."""
        cam = camera or self.camera
        out = np.empty((cam.vsize, cam.hsize, 3), dtype=np.uint8)
        _check(render_lib().rl_rtc_render_rgb8(self.device(), C.byref(cam), aa_samples, out.ctypes.data, None), allow_degenerate=True)
        return out

    def render_multi(self, aa_samples=1, camera=None, stats=None, allow_degenerate=False):
        """rl_rtc_render_multi: the whole frame over every GPU of init_multi."""
        cam = camera or self.camera
        out = np.empty((cam.vsize, cam.hsize, 3), dtype=np.float64)
        st = Stats()
        rc = render_lib().rl_rtc_render_multi(self.device(), C.byref(cam), aa_samples, out.ctypes.data, C.byref(st))
        _check(rc, allow_degenerate)
        if stats is not None:
            stats.update(st.as_dict())
            stats["rc"] = rc
        return out

    def render_device(self, d_ptr, aa_samples=1, camera=None, stream=0, row_first=0, row_step=1, stats=None):
        cam = camera or self.camera
        st = Stats() if stats is not None else None
        rc = render_lib().rl_rtc_render_device(self.device(), C.byref(cam), aa_samples, row_first, row_step, C.c_void_p(d_ptr),
                                               C.c_void_p(stream), C.byref(st) if st is not None else None)
        _check(rc)
        if stats is not None:
            stats.update(st.as_dict())


def rtc_camera(hsize, vsize, fov, frm, to, up) -> RtcCamera:  # Camera::new + view_transform
    c = RtcCamera()
    a = [np.ascontiguousarray(v, dtype=np.float64) for v in (frm, to, up)]
    if host_lib().rlh_rtc_camera_new(hsize, vsize, fov, a[0].ctypes.data, a[1].ctypes.data, a[2].ctypes.data, C.byref(c)) != 0:
        raise RuntimeError("rtc Camera::new: " + host_lib().rlh_last_error().decode())
    return c


def rtc_transformed(matrix4, child_kind, child_index):
    """Transformed::new(child, transform): one RTC_TRANSFORMED record (inverse + inverse-transpose by cofactors)."""
    rec = np.zeros(1, dtype=RTC_TRANSFORMED)
    m = np.ascontiguousarray(matrix4, dtype=np.float64).reshape(16)
    if host_lib().rlh_rtc_make_transformed(m.ctypes.data, rec.ctypes.data) != 0:
        raise RuntimeError("Matrix is not invertible.")
    rec["child"]["kind"], rec["child"]["index"] = child_kind, child_index
    return rec[0]


def canvas_ppm(rgb) -> str:  # draw/canvas.rs:50-97
    rgb = np.ascontiguousarray(rgb, dtype=np.float64)
    h, w = rgb.shape[:2]
    n = C.c_uint64()
    return _take_string(host_lib().rlh_rtc_canvas_ppm(rgb.ctypes.data, w, h, C.byref(n)), n)
