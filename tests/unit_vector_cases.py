"""The scenes tests/golden/make_unit_vectors.cpp builds from the reference's classes, restated as rt_scene_desc, and
the comparison of a (t, prim, normal) answer with the fixture rows (data: tests/golden/unit_vectors.npz)."""
import os

import numpy as np

from raytracer_amd import capi
from tests import scene_util

GOLDEN = os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden", "unit_vectors.npz")


def vectors():
    return np.load(GOLDEN)


def row(kind, *vals):
    g = np.zeros(capi.RT_GEOM_STRIDE)
    g[:len(vals)] = vals
    return kind, g


PRIMS = {
    "sphere": row(capi.RT_SPHERE, 0.5, -0.25, 3.0, 1.25),
    "triangle": row(capi.RT_TRIANGLE, -1, -0.5, 2, 1.5, -0.5, 2.5, 0.25, 1.5, 4),
    "cylinder_y": row(capi.RT_CYLINDER, 0.3, -0.5, 1.0, 0, 1, 0, 0.25, 1.0),
    "cylinder_x": row(capi.RT_CYLINDER, -1.0, 0.5, 2.0, 1, 0, 0, 0.5, 2.0),
    "cylinder_d": row(capi.RT_CYLINDER, 0.0, 0.0, 3.0, 1, 2, -0.5, 0.4, 1.5),  # the constructor normalizes the axis
}
SCENE_ORDER = ["sphere", "triangle", "cylinder_y", "cylinder_x", "cylinder_d"]  # material.ks = index in the driver


def one_prim_scene(name):
    kind, g = PRIMS[name]
    return scene_util.SyntheticScene([kind], [g], use_bvh=0)


def five_prim_scene(use_bvh):
    kinds = [PRIMS[n][0] for n in SCENE_ORDER]
    geom = [PRIMS[n][1] for n in SCENE_ORDER]
    return scene_util.SyntheticScene(kinds, geom, use_bvh=use_bvh)


def check_primitive(name, V, t, prim, nrm):
    """(t, prim, normal) of a one-primitive scene against <name>_out = hit, t, point xyz, normal xyz."""
    out = V[name + "_out"]
    hit = out[:, 0] > 0
    assert np.array_equal(prim >= 0, hit), f"{name}: hit / miss differs on {(hit != (prim >= 0)).sum()} rays"
    assert np.array_equal(t[hit], out[hit, 1]), f"{name}: t differs"
    assert np.array_equal(nrm[hit], out[hit, 5:8], equal_nan=True), f"{name}: normal differs"


def check_scene(key, V, t, prim, nrm):
    """against scene_*_out = prim (material tag) or -1, t, normal xyz."""
    out = V[key + "_out"]
    assert np.array_equal(prim, out[:, 0].astype(np.int32)), f"{key}: primitive differs on {(prim != out[:, 0]).sum()} rays"
    hit = prim >= 0
    assert np.array_equal(t[hit], out[hit, 1]), f"{key}: t differs"
    assert np.array_equal(nrm[hit], out[hit, 2:5], equal_nan=True), f"{key}: normal differs"
