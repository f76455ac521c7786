"""A/B timing: ref = persistent per-pixel kernel (RT_PIPELINE=mega), fast = wavefront pipeline; in separate contexts of one process (RT_TRAVERSAL is read at rt_create)."""
import os, sys, time
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import raytracer_amd as rt

def bench(r, sf, desc, P, n=5):
    r.upload(desc); r.set_camera(sf.camera); r.set_params(P)
    rgb, hit = r.render()
    ms = []
    for _ in range(n):
        r.render(want_hit=False); ms.append(r.stats().kernel_ms)
    st = r.stats()
    return rgb, hit, st, float(np.median(ms)), float(np.min(ms))

cases = [("clutter_500", 3840, 2160, 1), ("clutter_500", 1200, 800, 1), ("clutter_500", 1200, 800, 0), ("mirror_glass_cylinder", 1920, 1080, 1),
         ("binary_primitives", 3840, 2160, 1), ("pathtrace", 480, 270, 1), ("final", 480, 270, 1)]
ctx = {}
for mode in ["ref", "fast"]:
    os.environ["RT_PIPELINE"] = os.environ.get("AB_REF", "block") if mode == "ref" else os.environ.get("AB_FAST", "fused")
    ctx[mode] = rt.Renderer(0)
for name, w, h, bvh in cases:
    sf = rt.SceneFile(f"scenes/{name}.json", width=w, height=h)
    desc = sf.desc; desc.use_bvh = bvh
    P = sf.params
    if P.mode == 2: P.pixel_samples, P.light_samples = 16, 16
    out = {}
    for mode in ["ref", "fast"]:
        out[mode] = bench(ctx[mode], sf, desc, P)
    a, b = out["ref"], out["fast"]
    same = np.array_equal(a[0], b[0]) and (a[1] is None or np.array_equal(a[1], b[1])) and a[2].rays == b[2].rays
    print(f"{name:22s} {w}x{h} bvh={bvh}: rays={b[2].rays} identical={same}  ref {a[3]:.2f} ms ({a[2].rays/a[3]/1e3:.0f} Mrays/s)   fast {b[3]:.2f} ms ({b[2].rays/b[3]/1e3:.0f} Mrays/s)  x{a[3]/b[3]:.2f}  box/ray {a[2].box_tests/a[2].rays:.1f}->{b[2].box_tests/b[2].rays:.1f} prim/ray {a[2].prim_tests/a[2].rays:.2f}->{b[2].prim_tests/b[2].rays:.2f}", flush=True)
