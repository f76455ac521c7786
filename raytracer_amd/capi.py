"""ctypes mirror of include/raytracer_amd.h.

Everything here is a thin typed view of the C-ABI shared library
(raytracer_amd/csrc/libraytracer_amd.so).  There is no Python fallback: if the library (and,
for device calls, its HIP code object) is missing, loading raises.
"""
import ctypes as C
import os

_HERE = os.path.dirname(os.path.abspath(__file__))
LIB_PATH = os.environ.get("RT_AMD_LIB") or os.path.join(_HERE, "csrc", "libraytracer_amd.so")  # RT_AMD_LIB: A/B builds

RT_GEOM_STRIDE = 12
RT_SPHERE, RT_TRIANGLE, RT_CYLINDER = 0, 1, 2
RT_MODE_PHONG, RT_MODE_BINARY, RT_MODE_PATHTRACE = 0, 1, 2
RT_TONEMAP_NONE, RT_TONEMAP_REINHARD, RT_TONEMAP_WARD, RT_TONEMAP_UNCHARTED2 = 0, 1, 2, 3
RT_LIGHT_POINT, RT_LIGHT_AREA = 0, 1
RT_DEBUG_NO_JITTER = 1  # rt_params.debug_flags

d3 = C.c_double * 3


class Material(C.Structure):
    _fields_ = [
        ("ks", C.c_double), ("kd", C.c_double),
        ("reflectivity", C.c_double), ("refractive_index", C.c_double),
        ("diffuse", d3), ("specular", d3),
        ("specular_exponent", C.c_int32), ("is_reflective", C.c_int32),
        ("is_refractive", C.c_int32), ("texture", C.c_int32),
    ]


class Texture(C.Structure):
    _fields_ = [("width", C.c_int32), ("height", C.c_int32), ("rgb", C.POINTER(C.c_uint8))]


class Light(C.Structure):
    _fields_ = [
        ("type", C.c_int32), ("reserved", C.c_int32),
        ("intensity", d3), ("position", d3), ("normal", d3), ("u", d3), ("v", d3),
        ("width", C.c_double), ("height", C.c_double),
    ]


class SceneDesc(C.Structure):
    _fields_ = [
        ("n_prims", C.c_int32), ("n_materials", C.c_int32),
        ("n_textures", C.c_int32), ("n_lights", C.c_int32),
        ("prim_type", C.POINTER(C.c_int32)), ("prim_material", C.POINTER(C.c_int32)),
        ("prim_geom", C.POINTER(C.c_double)),
        ("materials", C.POINTER(Material)), ("textures", C.POINTER(Texture)),
        ("lights", C.POINTER(Light)),
        ("background", d3),
        ("use_bvh", C.c_int32), ("reserved", C.c_int32),
    ]


class Camera(C.Structure):
    _fields_ = [
        ("position", d3), ("lower_left", d3), ("horizontal", d3), ("vertical", d3),
        ("u", d3), ("v", d3), ("lens_radius", C.c_double),
    ]


class Params(C.Structure):
    _fields_ = [
        ("width", C.c_int32), ("height", C.c_int32),
        ("mode", C.c_int32), ("tonemap", C.c_int32),
        ("max_depth", C.c_int32), ("pixel_samples", C.c_int32),
        ("light_samples", C.c_int32), ("reserved0", C.c_int32),
        ("exposure", C.c_double), ("shadow_bias", C.c_double),
        ("seed", C.c_uint64),
        ("band_rows", C.c_int32), ("band_rank", C.c_int32), ("band_world", C.c_int32),
        ("sample_begin", C.c_int32), ("sample_end", C.c_int32), ("debug_flags", C.c_int32),
    ]


class Stats(C.Structure):
    _fields_ = [
        ("rays", C.c_uint64), ("box_tests", C.c_uint64), ("prim_tests", C.c_uint64),
        ("sphere_tests", C.c_uint64), ("triangle_tests", C.c_uint64), ("cylinder_tests", C.c_uint64),
        ("kernel_ms", C.c_double), ("upload_ms", C.c_double),
        ("n_bvh_nodes", C.c_int32), ("scene_in_lds", C.c_int32),
        ("rays_traversed", C.c_uint64),
        ("kernel_launches", C.c_int32),
        ("reserved", C.c_int32),
        ("unwalked_ops", C.c_uint64),
    ]


# every symbol include/raytracer_amd.h declares: name -> (restype, argtypes)
_P = C.POINTER
SYMBOLS = {
    "rt_last_error": (C.c_char_p, []),
    "rt_api_version": (C.c_int, []),
    "rt_scene_file_load": (C.c_int, [C.c_char_p, C.c_char_p, _P(C.c_void_p)]),
    "rt_scene_file_free": (None, [C.c_void_p]),
    "rt_scene_file_resize": (C.c_int, [C.c_void_p, C.c_int32, C.c_int32]),
    "rt_scene_file_desc": (_P(SceneDesc), [C.c_void_p]),
    "rt_scene_file_camera": (_P(Camera), [C.c_void_p]),
    "rt_scene_file_params": (_P(Params), [C.c_void_p]),
    "rt_camera_look_at": (C.c_int, [_P(Camera), d3, d3, d3, C.c_double, C.c_double, C.c_double, C.c_double]),
    "rt_write_ppm": (C.c_int, [C.c_char_p, C.c_void_p, C.c_int32, C.c_int32]),
    "rt_build_bvh": (C.c_int, [_P(SceneDesc), C.c_int32, C.c_void_p, C.c_void_p, C.c_void_p]),
    "rt_device_count": (C.c_int, []),
    "rt_create": (C.c_int, [_P(C.c_void_p), C.c_int]),
    "rt_destroy": (None, [C.c_void_p]),
    "rt_upload_scene": (C.c_int, [C.c_void_p, _P(SceneDesc)]),
    "rt_set_camera": (C.c_int, [C.c_void_p, _P(Camera)]),
    "rt_set_params": (C.c_int, [C.c_void_p, _P(Params)]),
    "rt_local_rows": (C.c_int64, [C.c_void_p]),
    "rt_render": (C.c_int, [C.c_void_p, C.c_void_p, C.c_void_p]),
    "rt_render_device": (C.c_int, [C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p]),
    "rt_pathtrace_accumulate_device": (C.c_int, [C.c_void_p, C.c_void_p, C.c_void_p]),
    "rt_pathtrace_resolve_device": (C.c_int, [C.c_void_p, C.c_void_p, C.c_void_p, C.c_int64]),
    "rt_render_u8": (C.c_int, [C.c_void_p, C.c_void_p, C.c_void_p]),
    "rt_write_ppm_u8": (C.c_int, [C.c_char_p, C.c_void_p, C.c_int32, C.c_int32]),
    "rt_render_device_u8": (C.c_int, [C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p]),
    "rt_render_device_u8_static": (C.c_int, [C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p]),
    "rt_set_workgroups": (C.c_int, [C.c_void_p, C.c_int32]),
    "rt_quantize_device": (C.c_int, [C.c_void_p, C.c_void_p, C.c_void_p, C.c_int64, C.c_void_p]),
    "rt_get_stats": (C.c_int, [C.c_void_p, _P(Stats)]),
    "rt_group_create": (C.c_int, [_P(C.c_void_p), _P(C.c_int), C.c_int]),
    "rt_group_destroy": (None, [C.c_void_p]),
    "rt_group_size": (C.c_int, [C.c_void_p]),
    "rt_group_upload_scene": (C.c_int, [C.c_void_p, _P(SceneDesc)]),
    "rt_group_set_camera": (C.c_int, [C.c_void_p, _P(Camera)]),
    "rt_group_set_params": (C.c_int, [C.c_void_p, _P(Params)]),
    "rt_group_render": (C.c_int, [C.c_void_p, C.c_void_p, C.c_void_p]),
    "rt_group_render_u8": (C.c_int, [C.c_void_p, C.c_void_p, C.c_void_p]),
    "rt_group_get_stats": (C.c_int, [C.c_void_p, _P(Stats)]),
    "rt_debug_trace_rays": (C.c_int, [C.c_void_p, C.c_int64, C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p]),
    "rt_debug_math": (C.c_int, [C.c_void_p, C.c_int, C.c_int64, C.c_void_p, C.c_void_p, C.c_void_p]),
    "rt_debug_optics": (C.c_int, [C.c_void_p, C.c_int, C.c_int64, C.c_void_p, C.c_void_p]),
}

_lib = None


class RtError(RuntimeError):
    pass


def load(path=None):
    """Loads libraytracer_amd.so and types every declared symbol.  Raises if anything is missing."""
    global _lib
    if _lib is not None and path is None:
        return _lib
    p = path or LIB_PATH
    # One HIP runtime per process: PyTorch-ROCm wheels bundle their own libamdhip64.  If this library
    # pulled in /opt/rocm's copy first, torch.cuda would later find "No HIP GPUs".  Importing torch
    # first lets the loader bind our NEEDED libamdhip64 to the copy torch already mapped.
    if os.environ.get("RT_NO_TORCH_PRELOAD") != "1":
        try:
            import torch  # noqa: F401
        except Exception:
            pass
    if not os.path.exists(p):
        raise RtError(
            f"{p} not found: build it with `make -C raytracer_amd/csrc` "
            "(or python -c 'import __graft_entry__ as g; g.build()')")
    lib = C.CDLL(p)
    for name, (res, args) in SYMBOLS.items():
        fn = getattr(lib, name)  # AttributeError if the symbol is not exported
        fn.restype = res
        fn.argtypes = args
    if path is None:
        _lib = lib
    return lib


def check(code, what="call"):
    if code < 0:
        msg = load().rt_last_error()
        raise RtError(f"{what} failed ({code}): {msg.decode() if msg else ''}")
    return code
