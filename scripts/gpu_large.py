"""Times scenes whose image does not fit in LDS (synthetic soups of N primitives), default pipeline."""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
os.chdir(ROOT)
import numpy as np
import raytracer_amd as rt
from tests import scene_util
from tests.test_gpu_parity import large_soup

r = rt.Renderer(0)
for n_sph, n_tri, n_cyl in [(300, 60, 20), (1500, 300, 100), (15000, 3000, 1000)]:
    types, geom, mats, pm, lights, cam = large_soup(7, n_sph, n_tri, n_cyl)
    s = scene_util.SyntheticScene(types, geom, materials=mats, lights=lights, prim_material=pm, use_bvh=1)
    P = scene_util.make_params(1920, 1080, max_depth=5)
    r.upload(s.desc); r.set_camera(cam); r.set_params(P)
    r.render(want_hit=False)
    ms = []
    for _ in range(3):
        r.render(want_hit=False); ms.append(r.stats().kernel_ms)
    st = r.stats()
    print(f"soup {len(types):6d} prims 1920x1080 lds={st.scene_in_lds} rays={st.rays} {np.median(ms):.3f} ms  "
          f"{st.rays/np.median(ms)/1e3:.0f} Mrays/s", flush=True)

# generated clutter scenes (scripts/gen_clutter.py), 4K
import importlib.util, json, tempfile
spec = importlib.util.spec_from_file_location("gen_clutter", os.path.join(ROOT, "scripts", "gen_clutter.py"))
gen = importlib.util.module_from_spec(spec); spec.loader.exec_module(gen)
for n in (500, 5000, 50000, 500000):
    with tempfile.NamedTemporaryFile("w", suffix=".json", delete=False) as f:
        json.dump(gen.clutter(n, seed=1, width=3840, height=2160), f)
    sf = rt.SceneFile(f.name)
    os.unlink(f.name)
    r.upload(sf.desc); r.set_camera(sf.camera); r.set_params(sf.params)
    r.render(want_hit=False)
    ms = []
    for _ in range(3):
        r.render(want_hit=False); ms.append(r.stats().kernel_ms)
    st = r.stats()
    print(f"clutter {n:7d} spheres 3840x2160 lds={st.scene_in_lds} upload={st.upload_ms:.0f} ms rays={st.rays} {np.median(ms):.3f} ms  "
          f"{st.rays/np.median(ms)/1e3:.0f} Mrays/s", flush=True)
