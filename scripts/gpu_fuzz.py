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
n_cases = n_bad = n_group = n_static = 0
kinds = {}
seed = seed0
one = os.environ.get("FUZZ_ONE") == "1"  # exactly one case (the first seed), reported whatever its outcome
while time.time() - t0 < budget or (one and n_cases == 0):
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
    textures = None
    if rng.random() < 0.3:  # a random texture on the first and the last material (getUV of every primitive class, getTextureColor)
        textures = [rng.integers(0, 256, (int(rng.integers(1, 9)), int(rng.integers(1, 9)), 3)).astype(np.uint8)]
        mats[0].texture = 0
        mats[3].texture = 0
    if os.environ.get("FUZZ_TEX_CONST") == "1" and textures is not None:  # (investigations: every texel the same colour)
        textures = [np.broadcast_to(textures[0][0, 0], textures[0].shape).copy()]
    if os.environ.get("FUZZ_NOTEX") == "1" and textures is not None:  # (investigations: the same scene without its texture)
        textures = None
        mats[0].texture = -1
        mats[3].texture = -1
    pm = rng.integers(0, len(mats), len(types))
    pt = mode == rt.RT_MODE_PATHTRACE
    # getUV on a cylinder's TOP cap is a coin toss of the hit point's last bit: v = clamp(y / height, 0, 1) is 1 or 1 - 1e-16
    # there, and getTextureColor maps the one to the texture's last row (v - floor(v) = 0 -> (1 - 0) h, clamped) and the other
    # to its first.  Library and oracle toss the same coin for the same point (Whitted frames: exact), but a path tracer's
    # points carry the last-bit differences of cos / sin from the first diffuse bounce on.  So path-traced cases keep
    # textures off their cylinders (strict criterion), except one in ten, which is held to a loose one.
    loose = False
    if pt and textures is not None:
        cyl_tex = (types == rt.capi.RT_CYLINDER) & ((pm == 0) | (pm == 3))
        if rng.random() < 0.1:
            loose = bool(cyl_tex.any())
        else:
            pm[cyl_tex] = 1 + (pm[cyl_tex] & 1)
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
    syn = scene_util.SyntheticScene(types, geom, materials=mats, lights=lights, use_bvh=int(rng.integers(0, 2)), prim_material=pm, textures=textures)
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
    if os.environ.get("FUZZ_DEPTH"):  # (investigations: the same case with another recursion limit)
        P.max_depth = int(os.environ["FUZZ_DEPTH"])
    what = "seed %d mode %d prims %d/%d/%d bvh %d %dx%d depth %d spp %d ls %d lights %d" % (
        seed, mode, n_sph, n_tri, n_cyl, syn.desc.use_bvh, w, h, P.max_depth, spp, ls, len(lights))
    try:
        r.upload(syn.desc); r.set_camera(cam); r.set_params(P)
        rgb, hit = r.render(want_hit=not pt)
        st = r.stats()
        orgb, ohit, cnt, _ = oracle_lib.render(syn.desc, cam, P, want_hit=not pt)
        if pt:
            dpx = (np.abs(rgb - orgb) / np.maximum(1.0, np.abs(orgb))).max(axis=2)  # (relative where a small, near light makes the radiance large)
            frac = float((dpx > 1e-9).mean())
            lim = 1e-2 if loose else 1e-3
            ok = (frac <= lim + 2.0 / (w * h) and abs(st.rays - cnt["rays"]) <= 1e-3 * cnt["rays"] + 64 and bool(np.isfinite(rgb).all()))
            detail = "frac %.2e largest %.2e rays %d / %d" % (frac, float(dpx.max()), st.rays, cnt["rays"])
        else:
            ok = bool(np.array_equal(hit, ohit)) and st.rays == cnt["rays"] and float(np.abs(rgb - orgb).max()) <= 1e-12
            detail = "hits equal %s rays %d / %d max diff %.2e" % (bool(np.array_equal(hit, ohit)), st.rays, cnt["rays"], float(np.abs(rgb - orgb).max()))
        if ok and not pt and P.max_depth > 0 and rng.random() < 0.1:  # the same frame out of the static workspace, as PPM bytes
            import torch
            buf = torch.full((h * w * 3,), 77, dtype=torch.uint8, device="cuda")
            r.render_device_u8_static(buf.data_ptr(), None, None, stream=torch.cuda.current_stream().cuda_stream)
            torch.cuda.synchronize()
            r.stats()
            n_static += 1
            ok = bool(np.array_equal(buf.cpu().numpy().reshape(h, w, 3), (255.999 * rgb).astype(np.int32).astype(np.uint8)))
            if not ok:
                detail = "rt_render_device_u8_static differs from the allocating entry point"
        if ok and not heavy and rng.random() < 0.1:  # the same frame from a group of 2-4 members on this device
            class _SF:  # (what Group.load reads)
                pass
            sfl = _SF(); sfl.camera = cam; sfl.desc = syn.desc; sfl.params = P
            g = rt.Group([0] * int(rng.integers(2, 5)))
            try:
                g.load(sfl, P)
                grgb, ghit = g.render(want_hit=not pt)
            finally:
                g.close()
            n_group += 1
            if pt:
                ok = float(np.abs(grgb - rgb).max()) < 1e-12
            else:
                ok = bool(np.array_equal(grgb, rgb)) and bool(np.array_equal(ghit, hit))
            if not ok:
                detail = "group frame differs from the one-context frame (max %.2e)" % float(np.abs(grgb - rgb).max())
    except Exception as e:  # a refused configuration is a finding too
        ok, detail = False, "raised %s" % e
    n_cases += 1
    kinds[mode] = kinds.get(mode, 0) + 1
    if not ok or one:
        n_bad += 0 if ok else 1
        print("FAIL" if not ok else "ok", what, "|", detail, "| textured %s" % (textures is not None), flush=True)
    if one:
        d = np.abs(rgb - orgb).max(axis=2)
        ys, xs = np.nonzero(d > 1e-9)
        for y, x in list(zip(ys, xs))[:3]:
            print("   pixel (%d, %d): library %s oracle %s" % (y, x, rgb[y, x].tolist(), orgb[y, x].tolist()))
        print("pixels off: %d of %d; largest %.3e; first few (y, x, diff): %s" % (len(ys), d.size, float(d.max()), [(int(y), int(x), float(d[y, x])) for y, x in list(zip(ys, xs))[:8]]))
        # getUV + getTextureColor of this scene's primitives at the hit points of random rays: library against oracle
        rr = np.random.default_rng(seed + 7)
        no = 400000
        oo = rr.uniform(-6, 6, (no, 3)); dd = rr.normal(size=(no, 3)); dd /= np.linalg.norm(dd, axis=1)[:, None]
        tt, pp, _ = r.trace_rays(oo, dd)
        keep = pp >= 0
        pts = oo[keep] + dd[keep] * tt[keep][:, None]
        pr = pp[keep]
        got = r.optics(2, np.column_stack([pr.astype(np.float64), pts]))
        want = oracle_lib.uv_texture(syn.desc, pr, points=pts)
        du = np.abs(got[:, 0:2] - want[:, 0:2]).max(axis=1)
        dc = np.abs(got[:, 2:5] - want[:, 2:5]).max(axis=1)
        bad = np.nonzero(dc > 0)[0]
        print("uv / texture at %d hit points: max |duv| %.2e, texture colour differs at %d points" % (len(pr), float(du.max()), len(bad)))
        for k in bad[:6]:
            print("   prim %d type %d point %s: uv got %s want %s colour got %s want %s" % (pr[k], types[pr[k]], pts[k], got[k, 0:2], want[k, 0:2], got[k, 2:5], want[k, 2:5]))
        print("primitive types:", dict(zip(*np.unique(types, return_counts=True))), "materials of prims:", dict(zip(*np.unique(pm, return_counts=True))))
        break
    seed += 1
    if time.time() - t_last > 30.0:  # (a long run that prints nothing looks hung to the box)
        t_last = time.time()
        print("... %d cases, %d failed, %.0f s" % (n_cases, n_bad, t_last - t0), flush=True)
print("%d cases (by mode %s; %d also through a group, %d through the static workspace) in %.0f s, %d failed; seeds %d..%d" % (n_cases, kinds, n_group, n_static, time.time() - t0, n_bad, seed0, seed - 1))
sys.exit(1 if n_bad else 0)
