"""The C-ABI library loads on a machine without a GPU and exports every symbol the header declares."""
import os
import re
import subprocess

import pytest

from raytracer_amd import capi

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
HEADER = os.path.join(ROOT, "include", "raytracer_amd.h")


def declared_functions():
    text = open(HEADER).read()
    text = re.sub(r"/\*.*?\*/", "", text, flags=re.S)
    return sorted(set(re.findall(r"\b(rt_[a-z0-9_]+)\s*\(", text)))


def test_header_and_bindings_agree():
    assert declared_functions() == sorted(capi.SYMBOLS)


def test_library_exports_every_declared_symbol():
    lib = capi.load()  # raises if a symbol is missing
    assert lib.rt_api_version() == 1
    out = subprocess.run(["nm", "-D", "--defined-only", capi.LIB_PATH], capture_output=True, text=True, check=True).stdout
    exported = set(re.findall(r"\bT (rt_[a-z0-9_]+)", out))
    assert set(declared_functions()) <= exported


def test_library_contains_gfx950_code_object():
    data = open(capi.LIB_PATH, "rb").read()
    assert b"gfx950" in data and b"render_kernel" in data


def test_struct_sizes_match_header_layout():
    import ctypes as C
    assert C.sizeof(capi.Material) == 4 * 8 + 6 * 8 + 4 * 4
    assert C.sizeof(capi.Light) == 8 + 15 * 8 + 16
    assert C.sizeof(capi.Camera) == 19 * 8
    assert C.sizeof(capi.Params) == 8 * 4 + 3 * 8 + 6 * 4
    assert C.sizeof(capi.SceneDesc) == 16 + 6 * 8 + 24 + 8


def test_no_cpu_fallback_without_device():
    import ctypes as C
    lib = capi.load()
    if lib.rt_device_count() > 0:
        pytest.skip("a GPU is present")
    h = C.c_void_p()
    rc = lib.rt_create(C.byref(h), 0)
    assert rc == -3 and not h.value  # RT_ERR_DEVICE: fails loudly, no fallback
    assert b"no CPU fallback" in lib.rt_last_error()
