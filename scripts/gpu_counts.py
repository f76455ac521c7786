"""Executed test counts of the production traversal (a -DRT_FBVH_COUNTS=1 build, RT_AMD_LIB) next to the
reference-order counts (RT_TRAVERSAL=walk) on one frame of a case: name w h [spp ls]."""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
os.chdir(ROOT)
import raytracer_amd as rt
name, w, h = sys.argv[1], int(sys.argv[2]), int(sys.argv[3])
spp, ls = (int(sys.argv[4]), int(sys.argv[5])) if len(sys.argv) > 5 else (4, 16)
sf = rt.SceneFile(f"scenes/{name}.json", width=w, height=h)
desc = sf.desc; desc.use_bvh = 1
P = sf.params
if P.mode == 2: P.pixel_samples, P.light_samples = spp, ls
for trav in ["walk", None]:
    if trav: os.environ["RT_TRAVERSAL"] = trav
    else: os.environ.pop("RT_TRAVERSAL", None)
    r = rt.Renderer(0)
    r.upload(desc); r.set_camera(sf.camera); r.set_params(P)
    r.render(want_hit=False)
    st = r.stats()
    print(f"{name} {trav or 'fbvh'}: rays {st.rays} box/ray {st.box_tests/st.rays:.3f} sphere/ray {st.sphere_tests/st.rays:.3f} "
          f"tri/ray {st.triangle_tests/st.rays:.3f} cyl/ray {st.cylinder_tests/st.rays:.3f} kernel {st.kernel_ms:.2f} ms")
    r.close()
