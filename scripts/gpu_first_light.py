"""First-light check on the GPU: device math exactness, ray parity, frame parity vs the oracle."""
import sys, os, time, hashlib
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import raytracer_amd as rt
from raytracer_amd import capi
from tests import oracle_lib

r = rt.Renderer(0)
rng = np.random.default_rng(1)
# --- math exactness
n = 2_000_000
a = rng.standard_normal(n) * np.exp(rng.uniform(-20, 20, n)); b = rng.standard_normal(n) * np.exp(rng.uniform(-20, 20, n))
print("div exact:", np.array_equal(r.math(0, a, b), a / b))
x = np.abs(a)
print("sqrt exact:", np.array_equal(r.math(1, x), np.sqrt(x)))
base = rng.uniform(0, 1, n); ex = rng.integers(1, 200, n).astype(np.float64)
pd = r.math(2, base, ex); pc = np.power(base, ex)
print("pow mismatches: %d of %d, max rel %.3g" % ((pd != pc).sum(), n, np.nanmax(np.abs(pd - pc) / np.maximum(np.abs(pc), 1e-300))))
for op, name, f in [(4, 'acos', np.arccos), (5, 'cos', np.cos), (6, 'sin', np.sin)]:
    xx = rng.uniform(-1, 1, n) if op == 4 else rng.uniform(0, 2 * np.pi, n)
    d = r.math(op, xx); c = f(xx)
    print(name, "mismatches", (d != c).sum(), "max abs", np.abs(d - c).max())

def run(name, w=None, h=None, bvh=None, **kw):
    sf = rt.SceneFile(f"scenes/{name}.json", width=w, height=h)
    desc = sf.desc
    if bvh is not None: desc.use_bvh = bvh
    P = sf.params
    for k, v in kw.items(): setattr(P, k, v)
    r.upload(desc); r.set_camera(sf.camera); r.set_params(P)
    t0 = time.time(); rgb, hit = r.render(); dt = time.time() - t0
    st = r.stats()
    t0 = time.time(); orgb, ohit, cnt, _ = oracle_lib.render(desc, sf.camera, P); odt = time.time() - t0
    diff = np.abs(rgb - orgb)
    rel = diff / np.maximum(np.abs(orgb), 1e-12)
    print(f"{name:24s} {P.width}x{P.height} bvh={desc.use_bvh} mode={P.mode}: hit_id equal={np.array_equal(hit, ohit) if P.mode!=2 else 'n/a'} "
          f"rgb exact={np.array_equal(rgb, orgb)} n_diff={(diff>0).sum()} max_abs={diff.max():.3g} max_rel={rel.max():.3g} "
          f"rays gpu={st.rays} cpu={cnt['rays']} box={st.box_tests}/{cnt['boxes']} prim={st.prim_tests}/{cnt['prims']} "
          f"kernel={st.kernel_ms:.2f}ms ({st.rays/st.kernel_ms/1e3:.1f} Mrays/s) lds={st.scene_in_lds} wall={dt*1e3:.0f}ms cpu={odt*1e3:.0f}ms")
    return rgb, orgb

for nm in ["binary_primitives", "blinn_phong", "mirror_glass_cylinder", "textured", "simple_phong", "tone_mapping"]:
    run(nm, 300, 200)
run("clutter_500", 300, 200, bvh=1)
run("clutter_500", 300, 200, bvh=0)
for nm in ["binary_primitives", "blinn_phong", "mirror_glass_cylinder"]:
    rgb, _ = run(nm)
    rt.write_ppm("/tmp/x.ppm", rgb); print("   sha256", hashlib.sha256(open("/tmp/x.ppm", "rb").read()).hexdigest())
rgb, _ = run("clutter_500", bvh=1)
rt.write_ppm("/tmp/x.ppm", rgb); print("   sha256", hashlib.sha256(open("/tmp/x.ppm", "rb").read()).hexdigest())
run("clutter_500", 3840, 2160, bvh=1)
# path trace
run("pathtrace", 150, 100)
run("final", 150, 100)
run("lens_sampling", 150, 100)
run("pathtrace", 300, 200, pixel_samples=16, light_samples=4)
