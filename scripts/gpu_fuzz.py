"""Randomised parity campaign on the GPU box (beyond what the test suite holds): random soups of every primitive class,
random materials, cameras, lights, sizes, depths and sample counts, each rendered by the library and by the oracle.
    python scripts/gpu_fuzz.py [seconds] [first seed]
Whitted / binary: hit ids and ray counts must be equal, colours within 1e-12.  Path trace: same ray count to 1e-3, at most
1e-3 of the pixels further than 1e-9 from the oracle (cos / sin / pow differ in the last bits and now and then flip a
Russian-roulette or hit / miss decision).  Prints one line per failing case and a summary; exit status 1 if any failed."""
import os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); os.chdir(ROOT)
import numpy as np
import raytracer_amd as rt
from tests import oracle_lib, scene_util

heavy = os.environ.get("FUZZ_HEAVY") == "1"  # fewer, larger path-traced frames with deep glass: full queues, many jobs per wave
budget = float(sys.argv[1]) if len(sys.argv) > 1 else 120.0
seed0 = int(sys.argv[2]) if len(sys.argv) > 2 else 1000
r = rt.Renderer(0)
t0 = t_last = time.time()
n_cases = n_bad = 0
kinds = {}
seed = seed0
while time.time() - t0 < budget:
    rng = np.random.default_rng(seed)
    mode = [rt.RT_MODE_PHONG, rt.RT_MODE_PATHTRACE, rt.RT_MODE_PATHTRACE, rt.RT_MODE_BINARY][int(rng.integers(0, 4))]
    if heavy:
        mode = rt.RT_MODE_PATHTRACE
    n_sph, n_tri, n_cyl = (int(rng.integers(0, k)) for k in (40, 30, 12))
    if n_sph + n_tri + n_cyl == 0:
        n_sph = 1
    types, geom = scene_util.random_prims(rng, n_sph, n_tri, n_cyl, extent=float(rng.uniform(1.5, 5.0)))
    if rng.random() < (0.7 if heavy else 0.3):  # something around everything: long ray trees
        shell = np.zeros((1, geom.shape[1])); shell[0, 3] = 9.0
        types = np.concatenate([np.array([rt.capi.RT_SPHERE], dtype=np.int32), types]); geom = np.concatenate([shell, geom])
    mats = [scene_util.default_material(diffuse=tuple(rng.uniform(0.1, 0.9, 3)), ks=float(rng.uniform(0, 0.6)), kd=float(rng.uniform(0.3, 1.0)),
                                        exp=int(rng.integers(1, 60))),
            scene_util.default_material(reflective=1, reflectivity=float(rng.uniform(0.1, 1.0))),
            scene_util.default_material(refractive=1, ior=float(rng.uniform(1.05, 2.0))),
            scene_util.default_material(diffuse=tuple(rng.uniform(0.1, 0.9, 3)), exp=int(rng.integers(1, 30)))]
    pm = rng.integers(0, len(mats), len(types))
    pt = mode == rt.RT_MODE_PATHTRACE
    if pt or rng.random() < 0.3:
        n = rng.normal(size=3); n /= np.linalg.norm(n)
        u = np.cross(n, rng.normal(size=3)); u /= np.linalg.norm(u)
        v = np.cross(n, u)
        lights = [scene_util.area_light(tuple(rng.uniform(-4, 4, 3) + np.array([0, 4, 0])), tuple(n), tuple(u), tuple(v),
                                        float(rng.uniform(0.2, 3.0)), float(rng.uniform(0.2, 3.0)))]
    else:
        lights = [scene_util.point_light(tuple(rng.uniform(-6, 6, 3))) for _ in range(int(rng.integers(1, 4)))]
    if pt and rng.random() < 0.15:
        lights = [scene_util.point_light(tuple(rng.uniform(-6, 6, 3)))]
    syn = scene_util.SyntheticScene(types, geom, materials=mats, lights=lights, use_bvh=int(rng.integers(0, 2)), prim_material=pm)
    w, h = int(rng.integers(8, 140)), int(rng.integers(8, 100))
    cam = scene_util.look_at(tuple(rng.uniform(-6, 6, 3)), tuple(rng.uniform(-1, 1, 3)), aspect=w / h,
                             aperture=float(rng.uniform(0, 0.3)) if (pt and rng.random() < 0.3) else 0.0)
    spp = int(rng.choice([1, 4, 5, 9, 16, 36]))
    ls = int(rng.choice([1, 2, 4, 8, 16, 16, 32, 3]))
    if heavy:
        w, h = int(rng.integers(150, 420)), int(rng.integers(100, 300))
        spp, ls = int(rng.choice([16, 64, 100])), int(rng.choice([16, 16, 4]))
    P = scene_util.make_params(w, h, mode=mode, max_depth=int(rng.integers(6, 17)) if heavy else int(rng.integers(0, 11)), pixel_samples=spp, light_samples=ls,
                               tonemap=int(rng.integers(0, 4)), exposure=float(rng.uniform(0.5, 2.0)), seed=int(rng.integers(0, 1 << 30)))
    what = "seed %d mode %d prims %d/%d/%d bvh %d %dx%d depth %d spp %d ls %d lights %d" % (
        seed, mode, n_sph, n_tri, n_cyl, syn.desc.use_bvh, w, h, P.max_depth, spp, ls, len(lights))
    try:
        r.upload(syn.desc); r.set_camera(cam); r.set_params(P)
        rgb, hit = r.render(want_hit=not pt)
        st = r.stats()
        orgb, ohit, cnt, _ = oracle_lib.render(syn.desc, cam, P, want_hit=not pt)
        if pt:
            frac = float((np.abs(rgb - orgb).max(axis=2) > 1e-9).mean())
            ok = frac <= 1e-3 + 2.0 / (w * h) and abs(st.rays - cnt["rays"]) <= 1e-3 * cnt["rays"] + 64 and bool(np.isfinite(rgb).all())
            detail = "frac %.2e rays %d / %d" % (frac, st.rays, cnt["rays"])
        else:
            ok = bool(np.array_equal(hit, ohit)) and st.rays == cnt["rays"] and float(np.abs(rgb - orgb).max()) <= 1e-12
            detail = "hits equal %s rays %d / %d max diff %.2e" % (bool(np.array_equal(hit, ohit)), st.rays, cnt["rays"], float(np.abs(rgb - orgb).max()))
    except Exception as e:  # a refused configuration is a finding too
        ok, detail = False, "raised %s" % e
    n_cases += 1
    kinds[mode] = kinds.get(mode, 0) + 1
    if not ok:
        n_bad += 1
        print("FAIL", what, "|", detail, flush=True)
    seed += 1
    if time.time() - t_last > 30.0:  # (a long run that prints nothing looks hung to the box)
        t_last = time.time()
        print("... %d cases, %d failed, %.0f s" % (n_cases, n_bad, t_last - t0), flush=True)
print("%d cases (by mode %s) in %.0f s, %d failed; seeds %d..%d" % (n_cases, kinds, time.time() - t0, n_bad, seed0, seed - 1))
sys.exit(1 if n_bad else 0)
