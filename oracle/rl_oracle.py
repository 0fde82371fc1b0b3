"""ctypes binding of the CPU oracle (oracle/librl_oracle.so).  TEST INFRASTRUCTURE ONLY: imported by
tests/, __graft_entry__.smoke() and bench.py's cpu_baseline leg — never by the product package."""
import ctypes as C
import os

import numpy as np

_HERE = os.path.dirname(os.path.abspath(__file__))
LIB = os.path.join(_HERE, "librl_oracle.so")
_lib = None


class Stats(C.Structure):
    _fields_ = [("rays", C.c_uint64), ("node_tests", C.c_uint64), ("sphere_tests", C.c_uint64),
                ("planar_tests", C.c_uint64), ("instance_enters", C.c_uint64), ("rng_words", C.c_uint64),
                ("flagged", C.c_uint64), ("kernel_ms", C.c_double)]

    def as_dict(self):
        return {k: getattr(self, k) for k, _ in self._fields_}


def lib():
    global _lib
    if _lib is None:
        if not os.path.exists(LIB):
            raise RuntimeError(f"{LIB} not built — run `make -C oracle`")
        L = C.CDLL(LIB)
        vp, u64, u32, i32, dbl = C.c_void_p, C.c_uint64, C.c_uint32, C.c_int, C.c_double
        L.rlo_rtiow_render_rows.argtypes = [vp, vp, u64, u32, u32, i32, vp, C.POINTER(Stats)]
        L.rlo_rtiow_render_pixels.argtypes = [vp, vp, u64, vp, vp, u32, i32, vp, C.POINTER(Stats)]
        L.rlo_bvh_build.argtypes = [vp, vp, u32, u32, vp, u32, C.POINTER(u32)]
        L.rlo_rtc_render_rows.argtypes = [vp, vp, u32, u32, u32, i32, vp, C.POINTER(Stats)]
        L.rlo_chacha_key.argtypes = [u64, vp]
        L.rlo_chacha_block.argtypes = [u64, u64, u64, vp]
        L.rlo_chacha_script.argtypes = [u64, vp, vp, u32, vp, C.POINTER(u64)]
        L.rlo_rtiow_pixel.argtypes = [vp, vp, u64, u32, u32, vp, C.POINTER(u64)]
        L.rlo_rtiow_hit.argtypes = [vp, vp, vp, dbl, dbl, dbl, vp]
        L.rlo_aabb_hit.argtypes = [vp, vp, vp, dbl, dbl]
        L.rlo_sphere_uv.argtypes = [vp, vp]
        L.rlo_perlin_noise.argtypes = [vp, vp]
        L.rlo_perlin_noise.restype = dbl
        L.rlo_perlin_turb.argtypes = [vp, vp, u32]
        L.rlo_perlin_turb.restype = dbl
        L.rlo_rtc_intersect.argtypes = [vp, vp, vp, vp, vp, vp, u32]
        L.rlo_rtc_color_at.argtypes = [vp, vp, vp, vp]
        L.rlo_rtc_lighting.argtypes = [vp, vp, vp, vp, vp, vp, dbl, vp]
        _lib = L
    return _lib


def hardware_threads():
    return lib().rlo_hardware_threads()


def _rows(height, row_first, row_step):
    return 0 if row_first >= height else (height - row_first + row_step - 1) // row_step


def rtiow_render(desc, cam, first_sample=0, row_first=0, row_step=1, threads=0, stats=None):
    """desc: pointer (int) to rl_rtiow_scene_desc; cam: ctypes rl_rtiow_camera. Returns [rows, W, 3] sums."""
    nrows = _rows(cam.image_height, row_first, row_step)
    out = np.empty((nrows, cam.image_width, 3), dtype=np.float64)
    st = Stats()
    rc = lib().rlo_rtiow_render_rows(desc, C.addressof(cam), first_sample, row_first, row_step, threads, out.ctypes.data, C.byref(st))
    if rc not in (0, -5):
        raise RuntimeError(f"oracle rtiow render rc={rc}")
    if stats is not None:
        stats.update(st.as_dict())
        stats["rc"] = rc
    return out


def rtiow_render_pixels(desc, cam, xs, ys, first_sample=0, threads=0, stats=None):
    """The per-pixel loop over an explicit pixel list, one task per pixel. Returns [n, 3] sums."""
    xs = np.ascontiguousarray(xs, dtype=np.uint32)
    ys = np.ascontiguousarray(ys, dtype=np.uint32)
    assert xs.shape == ys.shape and xs.ndim == 1
    out = np.empty((len(xs), 3), dtype=np.float64)
    st = Stats()
    rc = lib().rlo_rtiow_render_pixels(desc, C.addressof(cam), first_sample, xs.ctypes.data, ys.ctypes.data, len(xs), threads, out.ctypes.data, C.byref(st))
    if rc not in (0, -5):
        raise RuntimeError(f"oracle rtiow render_pixels rc={rc}")
    if stats is not None:
        stats.update(st.as_dict())
        stats["rc"] = rc
    return out


BVH_NODE = np.dtype([("bbox", "<f8", 6), ("n_children", "<u4"), ("reserved", "<u4"), ("child", [("kind", "<u4"), ("index", "<u4")], 2)])


def bvh_build(prim_boxes, prims, node_base=0):
    """Bvh::new restated (bvh.rs:22-77). prim_boxes [n, 6] f64, prims [n] (kind, index) -> rl_bvh_node records."""
    boxes = np.ascontiguousarray(prim_boxes, dtype=np.float64)
    prims = np.ascontiguousarray(prims)
    n = len(prims)
    out = np.zeros(max(1, 2 * n), dtype=BVH_NODE)
    cnt = C.c_uint32()
    rc = lib().rlo_bvh_build(boxes.ctypes.data, prims.ctypes.data, n, node_base, out.ctypes.data, len(out), C.byref(cnt))
    if rc != 0:
        raise RuntimeError(f"oracle bvh_build rc={rc}")
    return out[:cnt.value]


def rtc_render(desc, cam, aa=1, row_first=0, row_step=1, threads=0, stats=None):
    nrows = _rows(cam.vsize, row_first, row_step)
    out = np.empty((nrows, cam.hsize, 3), dtype=np.float64)
    st = Stats()
    rc = lib().rlo_rtc_render_rows(desc, C.addressof(cam), aa, row_first, row_step, threads, out.ctypes.data, C.byref(st))
    if rc not in (0, -5):
        raise RuntimeError(f"oracle rtc render rc={rc}")
    if stats is not None:
        stats.update(st.as_dict())
        stats["rc"] = rc
    return out


def chacha_key(seed):
    k = np.zeros(8, dtype=np.uint32)
    lib().rlo_chacha_key(seed, k.ctypes.data)
    return k


def chacha_block(seed, ctr, stream):
    b = np.zeros(16, dtype=np.uint32)
    lib().rlo_chacha_block(seed, ctr, stream, b.ctypes.data)
    return b


def chacha_script(seed, ops):
    """ops: list of ('f64',) | ('uniform',) | ('set_stream', s) | ('u64',). Returns (outputs, final_pos)."""
    code = {"f64": 0, "uniform": 1, "set_stream": 2, "u64": 3}
    o = np.array([code[op[0]] for op in ops], dtype=np.uint32)
    a = np.array([op[1] if len(op) > 1 else 0 for op in ops], dtype=np.uint64)
    out = np.zeros(len(ops), dtype=np.float64)
    pos = C.c_uint64()
    lib().rlo_chacha_script(seed, o.ctypes.data, a.ctypes.data, len(ops), out.ctypes.data, C.byref(pos))
    return out, pos.value


def rtiow_pixel(desc, cam, x, y, first_sample=0):
    out = np.zeros(3)
    words = C.c_uint64()
    lib().rlo_rtiow_pixel(desc, C.addressof(cam), first_sample, x, y, out.ctypes.data, C.byref(words))
    return out, words.value


def rtiow_hit(desc, o, d, time=0.0, tmin=0.0, tmax=float("inf")):
    o = np.ascontiguousarray(o, dtype=np.float64)
    d = np.ascontiguousarray(d, dtype=np.float64)
    out = np.zeros(11)
    if not lib().rlo_rtiow_hit(desc, o.ctypes.data, d.ctypes.data, time, tmin, tmax, out.ctypes.data):
        return None
    return dict(t=out[0], p=out[1:4].copy(), normal=out[4:7].copy(), front=bool(out[7]), u=out[8], v=out[9], mat=int(out[10]))


def sphere_uv(p):
    p = np.ascontiguousarray(p, dtype=np.float64)
    out = np.zeros(2)
    lib().rlo_sphere_uv(p.ctypes.data, out.ctypes.data)
    return float(out[0]), float(out[1])


def perlin_noise(perlin, p):
    """perlin: a 1-element array of api.PERLIN dtype (rl_perlin)."""
    p = np.ascontiguousarray(p, dtype=np.float64)
    return float(lib().rlo_perlin_noise(perlin.ctypes.data, p.ctypes.data))


def perlin_turb(perlin, p, depth=7):
    p = np.ascontiguousarray(p, dtype=np.float64)
    return float(lib().rlo_perlin_turb(perlin.ctypes.data, p.ctypes.data, depth))


def aabb_hit(bbox, o, d, tmin, tmax):
    b, o, d = (np.ascontiguousarray(v, dtype=np.float64) for v in (bbox, o, d))
    return bool(lib().rlo_aabb_hit(b.ctypes.data, o.ctypes.data, d.ctypes.data, tmin, tmax))


def rtc_intersect(desc, o, d, cap=1024):
    o = np.ascontiguousarray(o, dtype=np.float64)
    d = np.ascontiguousarray(d, dtype=np.float64)
    ts = np.zeros(cap)
    objs = np.zeros(cap, dtype=np.uint32)
    normals = np.zeros((cap, 3))
    n = lib().rlo_rtc_intersect(desc, o.ctypes.data, d.ctypes.data, ts.ctypes.data, objs.ctypes.data, normals.ctypes.data, cap)
    n = min(n, cap)
    return ts[:n], objs[:n], normals[:n]


def rtc_color_at(desc, o, d):
    o = np.ascontiguousarray(o, dtype=np.float64)
    d = np.ascontiguousarray(d, dtype=np.float64)
    out = np.zeros(3)
    lib().rlo_rtc_color_at(desc, o.ctypes.data, d.ctypes.data, out.ctypes.data)
    return out


def rtc_lighting(material_rec, point, light_pos, light_int, eyev, normalv, shadow_att):
    m = np.ascontiguousarray(material_rec)
    a = [np.ascontiguousarray(v, dtype=np.float64) for v in (point, light_pos, light_int, eyev, normalv)]
    out = np.zeros(3)
    lib().rlo_rtc_lighting(m.ctypes.data, a[0].ctypes.data, a[1].ctypes.data, a[2].ctypes.data, a[3].ctypes.data, a[4].ctypes.data, shadow_att, out.ctypes.data)
    return out
