"""ctypes wrapper of oracle/liboracle.so -- TEST INFRASTRUCTURE (the checker, never the product)."""
import ctypes as C
import os
import subprocess

import numpy as np

from raytracer_amd import capi

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
ORACLE_SO = os.path.join(ROOT, "oracle", "liboracle.so")
_lib = None


def load():
    global _lib
    if _lib is None:
        if not os.path.exists(ORACLE_SO):
            subprocess.check_call(["make", "-C", os.path.join(ROOT, "oracle")])
        lib = C.CDLL(ORACLE_SO)
        lib.oracle_render.restype = C.c_int
        lib.oracle_render.argtypes = [C.POINTER(capi.SceneDesc), C.POINTER(capi.Camera), C.POINTER(capi.Params),
                                      C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p, C.c_int]
        lib.oracle_trace_rays.restype = C.c_int
        lib.oracle_trace_rays.argtypes = [C.POINTER(capi.SceneDesc), C.c_int64, C.c_void_p, C.c_void_p,
                                          C.c_void_p, C.c_void_p, C.c_void_p]
        lib.oracle_box_hit.restype = C.c_int
        lib.oracle_box_hit.argtypes = [C.c_void_p, C.c_void_p, C.c_void_p]
        lib.oracle_bvh_leaf_order.restype = C.c_int
        lib.oracle_bvh_leaf_order.argtypes = [C.POINTER(capi.SceneDesc), C.c_void_p, C.c_void_p]
        lib.oracle_tonemap.restype = None
        lib.oracle_tonemap.argtypes = [C.c_void_p, C.c_int, C.c_void_p]
        lib.oracle_philox.restype = None
        lib.oracle_philox.argtypes = [C.c_void_p, C.c_void_p, C.c_void_p]
        lib.oracle_uniform.restype = C.c_double
        lib.oracle_uniform.argtypes = [C.c_uint64, C.c_uint32, C.c_uint32, C.c_uint32, C.c_uint32]
        lib.oracle_max_threads.restype = C.c_int
        lib.oracle_optics.restype = None
        lib.oracle_optics.argtypes = [C.c_int64, C.c_void_p, C.c_void_p]
        lib.oracle_direct_light.restype = C.c_int
        lib.oracle_direct_light.argtypes = [C.POINTER(capi.SceneDesc), C.POINTER(capi.Params), C.c_int64, C.c_void_p, C.c_void_p, C.c_void_p]
        lib.oracle_uv_texture.restype = C.c_int
        lib.oracle_uv_texture.argtypes = [C.POINTER(capi.SceneDesc), C.c_int64, C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p]
        _lib = lib
    return _lib


def render(desc, cam, params, want_hit=True, want_accum=False, threads=0):
    """Returns (rgb[H,W,3], hit_id[H,W] or None, counters dict, accum or None)."""
    lib = load()
    W, H = params.width, params.height
    rgb = np.zeros((H, W, 3), dtype=np.float64)
    hit = np.zeros((H, W), dtype=np.int32) if want_hit else None
    acc = np.zeros((H, W, 3), dtype=np.float64) if want_accum else None
    cnt = np.zeros(3, dtype=np.uint64)
    rc = lib.oracle_render(C.byref(desc), C.byref(cam), C.byref(params), rgb.ctypes.data,
                           hit.ctypes.data if want_hit else None, cnt.ctypes.data,
                           acc.ctypes.data if want_accum else None, threads)
    assert rc == 0
    return rgb, hit, {"rays": int(cnt[0]), "boxes": int(cnt[1]), "prims": int(cnt[2])}, acc


def trace_rays(desc, origins, dirs):
    lib = load()
    o = np.ascontiguousarray(origins, dtype=np.float64)
    d = np.ascontiguousarray(dirs, dtype=np.float64)
    n = o.shape[0]
    t = np.zeros(n, dtype=np.float64)
    prim = np.zeros(n, dtype=np.int32)
    nrm = np.zeros((n, 3), dtype=np.float64)
    lib.oracle_trace_rays(C.byref(desc), n, o.ctypes.data, d.ctypes.data, t.ctypes.data, prim.ctypes.data,
                          nrm.ctypes.data)
    return t, prim, nrm


def optics(rows):
    """rows [n, 8] = I, N, eta_t, eta_i -> [n, 9] = refract, transmits, reflect, fresnel, fresnelReflectance."""
    a = np.ascontiguousarray(rows, dtype=np.float64)
    out = np.zeros((a.shape[0], 9), dtype=np.float64)
    load().oracle_optics(a.shape[0], a.ctypes.data, out.ctypes.data)
    return out


def tonemap(colours):
    """colours [n, 3] -> [n, 12]: toneMap for NONE, REINHARD, WARD, UNCHARTED2."""
    a = np.ascontiguousarray(colours, dtype=np.float64)
    out = np.zeros((a.shape[0], 12), dtype=np.float64)
    lib = load()
    for k in range(a.shape[0]):
        for tm in range(4):
            lib.oracle_tonemap(a[k].ctypes.data, tm, out[k, 3 * tm:3 * tm + 3].ctypes.data)
    return out


def box_hit(box_lo_hi, origin, direction):
    b = np.ascontiguousarray(box_lo_hi, dtype=np.float64)
    o = np.ascontiguousarray(origin, dtype=np.float64)
    d = np.ascontiguousarray(direction, dtype=np.float64)
    return bool(load().oracle_box_hit(b.ctypes.data, o.ctypes.data, d.ctypes.data))


def direct_light(desc, params, origins, dirs):
    """estimateDirectLight at the closest hits of the rays: [n, 4] = (hit, direct xyz)."""
    o = np.ascontiguousarray(origins, dtype=np.float64)
    d = np.ascontiguousarray(dirs, dtype=np.float64)
    out = np.zeros((o.shape[0], 4), dtype=np.float64)
    assert load().oracle_direct_light(C.byref(desc), C.byref(params), o.shape[0], o.ctypes.data, d.ctypes.data, out.ctypes.data) == 0
    return out


def uv_texture(desc, prim, points=None, uv=None):
    """getUV + getTextureColor: [n, 5] = (u, v, r, g, b); uv given: the texture lookup only."""
    prim = np.ascontiguousarray(prim, dtype=np.int32)
    pts = np.ascontiguousarray(points, dtype=np.float64) if points is not None else None
    uvs = np.ascontiguousarray(uv, dtype=np.float64) if uv is not None else None
    out = np.zeros((prim.shape[0], 5), dtype=np.float64)
    assert load().oracle_uv_texture(C.byref(desc), prim.shape[0], prim.ctypes.data, pts.ctypes.data if pts is not None else None,
                                    uvs.ctypes.data if uvs is not None else None, out.ctypes.data) == 0
    return out
