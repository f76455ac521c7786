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


def pt_scene(V, use_bvh):
    """The mirror / glass scene under a point light of the fixture's pt_* keys (emitted by the driver as data): scene,
    camera, params -- pixelSamples 1 and RT_DEBUG_NO_JITTER make the camera rays render()'s, which the driver traced."""
    import raytracer_amd as rt
    prims, mats, setup = V["pt_prims"], V["pt_materials"], V["pt_setup"][0]
    kinds = prims[:, 0].astype(np.int32)
    geom = np.zeros((len(prims), capi.RT_GEOM_STRIDE))
    geom[:, :12] = prims[:, 2:14]
    materials = [scene_util.default_material(diffuse=tuple(m[7:10]), specular=tuple(m[10:13]), ks=m[0], kd=m[1], exp=int(m[2]),
                                             reflective=int(m[3]), reflectivity=m[4], refractive=int(m[5]), ior=m[6]) for m in mats]
    W, H, max_depth, fov, focus = int(setup[0]), int(setup[1]), int(setup[2]), setup[3], setup[4]
    light = scene_util.point_light(tuple(setup[11:14]), tuple(setup[14:17]))
    scene = scene_util.SyntheticScene(kinds, geom, materials=materials, lights=[light], background=tuple(setup[17:20]), use_bvh=use_bvh,
                                      prim_material=prims[:, 1].astype(np.int32))
    cam = scene_util.look_at(tuple(setup[5:8]), tuple(setup[8:11]), fov=fov, aspect=W / H, aperture=0.0, focus=focus)
    P = scene_util.make_params(W, H, mode=rt.RT_MODE_PATHTRACE, max_depth=max_depth, pixel_samples=1, light_samples=4)
    P.debug_flags = capi.RT_DEBUG_NO_JITTER
    return scene, cam, P


def camera_rays(cam, W, H):
    """render()'s rays (RayTracer.cpp:217-220 -> Camera.cpp:33-49, no lens): origins, directions as getRay hands them to the Ray constructor."""
    i, j = np.meshgrid(np.arange(W), np.arange(H))
    u, v = 1.0 - i / (W - 1), j / (H - 1)
    ll, hor, ver, pos = (np.array(list(x)) for x in (cam.lower_left, cam.horizontal, cam.vertical, cam.position))
    d = ((ll + u[..., None] * hor) + v[..., None] * ver) - pos   # Camera.cpp:44-46, no lens offset
    d = d / np.sqrt(d[..., 0:1] * d[..., 0:1] + d[..., 1:2] * d[..., 1:2] + d[..., 2:3] * d[..., 2:3])  # :48 normalizes; the Ray constructor will again
    return np.broadcast_to(pos, d.shape).reshape(-1, 3).copy(), d.reshape(-1, 3)


def textured_five_prim_scene():
    """five_prim_scene with the reference's checkerboard texture on every primitive (texture/checkerboard.ppm)."""
    tex = scene_util.load_p6(os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "texture", "checkerboard.ppm"))
    m = scene_util.default_material(diffuse=(0.1, 0.2, 0.3))
    m.texture = 0
    kinds = [PRIMS[n][0] for n in SCENE_ORDER]
    geom = [PRIMS[n][1] for n in SCENE_ORDER]
    return scene_util.SyntheticScene(kinds, geom, materials=[m], use_bvh=0, textures=[tex])
