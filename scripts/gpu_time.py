"""Times a few frames with the library RT_AMD_LIB names (default: the in-tree one)."""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
os.chdir(ROOT)
import numpy as np
import raytracer_amd as rt
cases = [("binary_primitives", 3840, 2160), ("blinn_phong", 3840, 2160), ("mirror_glass_cylinder", 3840, 2160), ("clutter_500", 3840, 2160)]
r = rt.Renderer(0)
for name, w, h in cases:
    sf = rt.SceneFile(f"scenes/{name}.json", width=w, height=h)
    desc = sf.desc; desc.use_bvh = 1
    r.upload(desc); r.set_camera(sf.camera); r.set_params(sf.params)
    r.render(want_hit=False)
    ms = []
    for _ in range(5):
        r.render(want_hit=False); ms.append(r.stats().kernel_ms)
    st = r.stats()
    print(f"{os.environ.get('RT_AMD_LIB','default')[-24:]:24s} {name:22s} {w}x{h} rays={st.rays} {np.median(ms):.3f} ms  {st.rays/np.median(ms)/1e3:.0f} Mrays/s", flush=True)
# path trace passes and the blocking host-buffer entry point (PCIe-inclusive)
import time
for name, w, h in [("pathtrace", 1920, 1080), ("final", 960, 540)]:
    sf = rt.SceneFile(f"scenes/{name}.json", width=w, height=h)
    P = sf.params; P.pixel_samples, P.light_samples = 16, 16
    r.upload(sf.desc); r.set_camera(sf.camera); r.set_params(P)
    r.render(want_hit=False)
    ms = []
    for _ in range(3):
        r.render(want_hit=False); ms.append(r.stats().kernel_ms)
    st = r.stats()
    print(f"{name:22s} {w}x{h} 16 strata rays={st.rays} {np.median(ms):.2f} ms  {st.rays/np.median(ms)/1e3:.0f} Mrays/s  {16/np.median(ms)*1e3:.0f} spp/s", flush=True)
sf = rt.SceneFile("scenes/clutter_500.json", width=3840, height=2160)
desc = sf.desc; desc.use_bvh = 1
r.upload(desc); r.set_camera(sf.camera); r.set_params(sf.params)
r.render(want_hit=False)
t0 = time.perf_counter()
for _ in range(3): r.render(want_hit=False)
dt = (time.perf_counter() - t0) / 3
print(f"rt_render (blocking, host buffer) clutter_500 4K: {dt*1e3:.1f} ms per frame = {r.stats().rays/dt/1e6:.0f} Mrays/s PCIe-inclusive")
