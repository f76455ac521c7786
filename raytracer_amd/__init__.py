"""raytracer_amd -- host-side mirror of the reference's render seam over the C-ABI.

The reference's seam is `RayTracer(scene, camera, W, H)` + setters + `render()` /
`renderPathTrace()` (/root/reference/src/RayTracer.cpp:102-149).  `SceneFile` is what main()
builds before the seam (JSON ingest), `Renderer` is the object behind it.  All compute happens in
raytracer_amd/csrc/libraytracer_amd.so on a gfx950 GPU; nothing here has a CPU fallback.
"""
import ctypes as C

import numpy as np

from . import capi
from .capi import (RT_MODE_BINARY, RT_MODE_PATHTRACE, RT_MODE_PHONG, RT_TONEMAP_NONE,  # noqa: F401
                   RT_TONEMAP_REINHARD, RT_TONEMAP_UNCHARTED2, RT_TONEMAP_WARD, RtError)

__all__ = ["SceneFile", "Renderer", "Group", "capi", "RtError", "copy_params"]


def copy_params(p):
    q = capi.Params()
    C.memmove(C.byref(q), C.byref(p), C.sizeof(capi.Params))
    return q


class SceneFile:
    """parseSceneSettings / parseCamera / parseLights / parseShapes (RayTracer.cpp:817-1035)."""

    def __init__(self, json_path, tonemap=None, width=None, height=None):
        self._lib = capi.load()
        self._h = C.c_void_p()
        capi.check(self._lib.rt_scene_file_load(str(json_path).encode(),
                                                tonemap.encode() if tonemap else None, C.byref(self._h)),
                   f"rt_scene_file_load({json_path})")
        if width and height:
            self.resize(width, height)

    def resize(self, width, height):
        capi.check(self._lib.rt_scene_file_resize(self._h, width, height), "rt_scene_file_resize")

    @property
    def desc(self):
        return self._lib.rt_scene_file_desc(self._h).contents

    @property
    def camera(self):
        return self._lib.rt_scene_file_camera(self._h).contents

    @property
    def params(self):
        """A COPY of the parsed params (edit freely)."""
        return copy_params(self._lib.rt_scene_file_params(self._h).contents)

    def close(self):
        if self._h:
            self._lib.rt_scene_file_free(self._h)
            self._h = C.c_void_p()

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass


class Renderer:
    """One GPU's renderer: RayTracer's setters and render()/renderPathTrace() over the C-ABI."""

    def __init__(self, device=0):
        self._lib = capi.load()
        self._h = C.c_void_p()
        capi.check(self._lib.rt_create(C.byref(self._h), device), "rt_create")
        self.params = None

    def upload(self, desc):
        capi.check(self._lib.rt_upload_scene(self._h, C.byref(desc)), "rt_upload_scene")

    def set_camera(self, cam):
        capi.check(self._lib.rt_set_camera(self._h, C.byref(cam)), "rt_set_camera")

    def set_params(self, params):
        capi.check(self._lib.rt_set_params(self._h, C.byref(params)), "rt_set_params")
        self.params = copy_params(params)

    def load(self, scene_file, params=None):
        self.upload(scene_file.desc)
        self.set_camera(scene_file.camera)
        self.set_params(params if params is not None else scene_file.params)

    @property
    def local_rows(self):
        return int(capi.check(self._lib.rt_local_rows(self._h), "rt_local_rows"))

    def render(self, want_hit=True):
        """Blocking render into host arrays: (rgb[rows, W, 3] float64, hit_id[rows, W] int32 or None)."""
        rows, W = self.local_rows, self.params.width
        rgb = np.empty((rows, W, 3), dtype=np.float64)
        hit = np.empty((rows, W), dtype=np.int32) if want_hit else None
        capi.check(self._lib.rt_render(self._h, rgb.ctypes.data, hit.ctypes.data if want_hit else None), "rt_render")
        return rgb, hit

    def render_u8(self, want_hit=False):
        """Blocking render of the PPM payload: (uint8 [rows, W, 3] = int(255.999 * c), hit_id or None)."""
        rows, W = self.local_rows, self.params.width
        u8 = np.empty((rows, W, 3), dtype=np.uint8)
        hit = np.empty((rows, W), dtype=np.int32) if want_hit else None
        capi.check(self._lib.rt_render_u8(self._h, u8.ctypes.data, hit.ctypes.data if want_hit else None), "rt_render_u8")
        return u8, hit

    def render_device(self, rgb_ptr, hit_ptr=None, stream=None):
        capi.check(self._lib.rt_render_device(self._h, stream, rgb_ptr, hit_ptr), "rt_render_device")

    def render_device_u8(self, rgb8_ptr, rgb_ptr=None, hit_ptr=None, stream=None):
        """Whitted / binary frame as PPM bytes (and, optionally, the fp64 frame as well)."""
        capi.check(self._lib.rt_render_device_u8(self._h, stream, rgb8_ptr, rgb_ptr, hit_ptr), "rt_render_device_u8")

    def render_device_u8_static(self, rgb8_ptr, rgb_ptr=None, hit_ptr=None, stream=None):
        """rt_render_device_u8 out of the library's static workspace: no allocation (small frames only)."""
        capi.check(self._lib.rt_render_device_u8_static(self._h, stream, rgb8_ptr, rgb_ptr, hit_ptr), "rt_render_device_u8_static")

    def accumulate_device(self, accum_ptr, stream=None):
        capi.check(self._lib.rt_pathtrace_accumulate_device(self._h, stream, accum_ptr),
                   "rt_pathtrace_accumulate_device")

    def resolve_device(self, accum_ptr, n_pixels, stream=None):
        capi.check(self._lib.rt_pathtrace_resolve_device(self._h, stream, accum_ptr, n_pixels),
                   "rt_pathtrace_resolve_device")

    def set_workgroups(self, n):
        """Caps the persistent kernels' grid (0 = fill the chip): for several contexts rendering at once."""
        capi.check(self._lib.rt_set_workgroups(self._h, int(n)), "rt_set_workgroups")

    def quantize_device(self, rgb_ptr, n_values, out_ptr, stream=None):
        """int(255.999 * c) per component (the PPM payload) from one device buffer into another."""
        capi.check(self._lib.rt_quantize_device(self._h, stream, rgb_ptr, n_values, out_ptr), "rt_quantize_device")

    def stats(self):
        st = capi.Stats()
        capi.check(self._lib.rt_get_stats(self._h, C.byref(st)), "rt_get_stats")
        return st

    def trace_rays(self, origins, dirs):
        o = np.ascontiguousarray(origins, dtype=np.float64)
        d = np.ascontiguousarray(dirs, dtype=np.float64)
        n = o.shape[0]
        t = np.empty(n, dtype=np.float64)
        prim = np.empty(n, dtype=np.int32)
        nrm = np.empty((n, 3), dtype=np.float64)
        capi.check(self._lib.rt_debug_trace_rays(self._h, n, o.ctypes.data, d.ctypes.data, t.ctypes.data,
                                                 prim.ctypes.data, nrm.ctypes.data), "rt_debug_trace_rays")
        return t, prim, nrm

    def math(self, op, a, b=None):
        a = np.ascontiguousarray(a, dtype=np.float64)
        bb = np.ascontiguousarray(b, dtype=np.float64) if b is not None else None
        out = np.empty_like(a)
        capi.check(self._lib.rt_debug_math(self._h, op, a.size, a.ctypes.data,
                                           bb.ctypes.data if bb is not None else None, out.ctypes.data),
                   "rt_debug_math")
        return out

    def optics(self, op, rows):
        """rt_debug_optics: op 0 rows [n, 8] -> [n, 9], op 1 colours [n, 3] -> [n, 12], op 2 / 3 rows [n, 4] -> [n, 5]."""
        a = np.ascontiguousarray(rows, dtype=np.float64)
        out = np.empty((a.shape[0], {0: 9, 1: 12}.get(op, 5)), dtype=np.float64)
        capi.check(self._lib.rt_debug_optics(self._h, op, a.shape[0], a.ctypes.data, out.ctypes.data), "rt_debug_optics")
        return out

    def close(self):
        if self._h:
            self._lib.rt_destroy(self._h)
            self._h = C.c_void_p()

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass


class Group:
    """Several GPUs of one node behind the seam (rt_group_*): bands / strata ranges, assembled on devices[0]."""

    def __init__(self, devices):
        self._lib = capi.load()
        self._h = C.c_void_p()
        arr = (C.c_int * len(devices))(*devices)
        capi.check(self._lib.rt_group_create(C.byref(self._h), arr, len(devices)), "rt_group_create")
        self.params = None

    def load(self, scene_file, params=None, desc=None):
        p = params if params is not None else scene_file.params
        capi.check(self._lib.rt_group_upload_scene(self._h, C.byref(desc if desc is not None else scene_file.desc)), "rt_group_upload_scene")
        capi.check(self._lib.rt_group_set_camera(self._h, C.byref(scene_file.camera)), "rt_group_set_camera")
        capi.check(self._lib.rt_group_set_params(self._h, C.byref(p)), "rt_group_set_params")
        self.params = copy_params(p)

    def render(self, want_hit=True):
        H, W = self.params.height, self.params.width
        rgb = np.empty((H, W, 3), dtype=np.float64)
        hit = np.empty((H, W), dtype=np.int32) if want_hit else None
        capi.check(self._lib.rt_group_render(self._h, rgb.ctypes.data, hit.ctypes.data if want_hit else None), "rt_group_render")
        return rgb, hit

    def render_u8(self, want_hit=False):
        H, W = self.params.height, self.params.width
        u8 = np.empty((H, W, 3), dtype=np.uint8)
        hit = np.empty((H, W), dtype=np.int32) if want_hit else None
        capi.check(self._lib.rt_group_render_u8(self._h, u8.ctypes.data, hit.ctypes.data if want_hit else None), "rt_group_render_u8")
        return u8, hit

    def stats(self):
        st = capi.Stats()
        capi.check(self._lib.rt_group_get_stats(self._h, C.byref(st)), "rt_group_get_stats")
        return st

    def close(self):
        if self._h:
            self._lib.rt_group_destroy(self._h)
            self._h = C.c_void_p()

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass


def write_ppm(path, rgb):
    """RayTracer::writeImageToPPM (RayTracer.cpp:352-367)."""
    rgb = np.ascontiguousarray(rgb, dtype=np.float64)
    H, W = rgb.shape[0], rgb.shape[1]
    capi.check(capi.load().rt_write_ppm(str(path).encode(), rgb.ctypes.data, W, H), "rt_write_ppm")


def write_ppm_u8(path, rgb8):
    """The same file from PPM payload bytes (Renderer.render_u8)."""
    rgb8 = np.ascontiguousarray(rgb8, dtype=np.uint8)
    H, W = rgb8.shape[0], rgb8.shape[1]
    capi.check(capi.load().rt_write_ppm_u8(str(path).encode(), rgb8.ctypes.data, W, H), "rt_write_ppm_u8")
