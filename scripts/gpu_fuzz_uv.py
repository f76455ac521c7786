"""Investigation: closest hits and getUV / getTextureColor of second- and third-generation rays (rays that start on surfaces,
as the path tracer's do), library against oracle, on the scene of one gpu_fuzz.py seed.
    python scripts/gpu_fuzz_uv.py <seed> [n_rays]"""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); os.chdir(ROOT)
import numpy as np
import raytracer_amd as rt
from tests import oracle_lib, scene_util

seed = int(sys.argv[1]); n = int(sys.argv[2]) if len(sys.argv) > 2 else 300000
# the scene exactly as gpu_fuzz.py builds it for this seed (same draws in the same order)
rng = np.random.default_rng(seed)
mode = int(rng.integers(0, 4))
n_sph, n_tri, n_cyl = (int(rng.integers(0, k)) for k in (40, 30, 12))
if n_sph + n_tri + n_cyl == 0:
    n_sph = 1
types, geom = scene_util.random_prims(rng, n_sph, n_tri, n_cyl, extent=float(rng.uniform(1.5, 5.0)))
if rng.random() < 0.3:
    shell = np.zeros((1, geom.shape[1])); shell[0, 3] = 9.0
    types = np.concatenate([np.array([rt.capi.RT_SPHERE], dtype=np.int32), types]); geom = np.concatenate([shell, geom])
mats = [scene_util.default_material(diffuse=tuple(rng.uniform(0.1, 0.9, 3)), ks=float(rng.uniform(0, 0.6)), kd=float(rng.uniform(0.3, 1.0)), exp=int(rng.integers(1, 60))),
        scene_util.default_material(reflective=1, reflectivity=float(rng.uniform(0.1, 1.0))),
        scene_util.default_material(refractive=1, ior=float(rng.uniform(1.05, 2.0))),
        scene_util.default_material(diffuse=tuple(rng.uniform(0.1, 0.9, 3)), exp=int(rng.integers(1, 30)))]
textures = None
if rng.random() < 0.3:
    textures = [rng.integers(0, 256, (int(rng.integers(1, 9)), int(rng.integers(1, 9)), 3)).astype(np.uint8)]
    mats[0].texture = 0
    mats[3].texture = 0
pm = rng.integers(0, len(mats), len(types))
print("scene of seed %d: %d prims, textured %s, texture %s" % (seed, len(types), textures is not None, None if textures is None else textures[0].shape[:2]))
syn = scene_util.SyntheticScene(types, geom, materials=mats, lights=[scene_util.point_light((0, 5, 0))], use_bvh=int(rng.integers(0, 2)), prim_material=pm, textures=textures)
r = rt.Renderer(0)
r.upload(syn.desc)
rr = np.random.default_rng(seed + 11)
o = rr.uniform(-6, 6, (n, 3)); d = rr.normal(size=(n, 3)); d /= np.linalg.norm(d, axis=1)[:, None]
for gen in range(4):
    t, p, nrm = r.trace_rays(o, d)
    ot, op, onrm = oracle_lib.trace_rays(syn.desc, o, d)
    same_hit = np.array_equal(p, op) and np.array_equal(t[p >= 0], ot[op >= 0])
    keep = p >= 0
    pts = o[keep] + d[keep] * t[keep][:, None]
    pr = p[keep]
    got = r.optics(2, np.column_stack([pr.astype(np.float64), pts]))
    want = oracle_lib.uv_texture(syn.desc, pr, points=pts)
    bad = np.nonzero(np.abs(got[:, 2:5] - want[:, 2:5]).max(axis=1) > 0)[0]
    print("generation %d: %d rays, %d hits, hits and t identical %s, normals max diff %.1e, max |duv| %.2e, colour differs at %d" % (
        gen, len(o), int(keep.sum()), same_hit, float(np.abs(nrm - onrm).max()), float(np.abs(got[:, 0:2] - want[:, 0:2]).max()), len(bad)))
    for k in bad[:5]:
        print("   prim %d type %d point %s uv got %s want %s" % (pr[k], types[pr[k]], pts[k], got[k, 0:2], want[k, 0:2]))
    # next generation: from the hit points, biased along the normal, random directions in both hemispheres
    nk = nrm[keep]
    nd = rr.normal(size=pts.shape); nd /= np.linalg.norm(nd, axis=1)[:, None]
    side = np.sign(np.einsum("ij,ij->i", nd, nk))[:, None]
    o = pts + nk * side * 1e-4
    d = nd
