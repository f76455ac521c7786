"""Renders one case a few times (for rocprofv3 --kernel-trace)."""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
os.chdir(ROOT)
import raytracer_amd as rt
name, w, h, n = sys.argv[1], int(sys.argv[2]), int(sys.argv[3]), int(sys.argv[4])
sf = rt.SceneFile(f"scenes/{name}.json", width=w, height=h)
desc = sf.desc; desc.use_bvh = 1
P = sf.params
spp = 16
for a in sys.argv[5:]:
    if a.startswith("spp"): spp = int(a[3:])
if P.mode == 2: P.pixel_samples, P.light_samples = spp, 16
r = rt.Renderer(0)
r.upload(desc); r.set_camera(sf.camera); r.set_params(P)
u8 = len(sys.argv) > 5 and sys.argv[5] == "u8"  # PPM bytes straight from the kernels, as bench.py renders
if u8:
    import torch
    out = torch.zeros(h * w * 3, dtype=torch.uint8, device="cuda")
for _ in range(n):
    if u8:
        r.render_device_u8(out.data_ptr(), None, None)
        torch.cuda.synchronize()
    else:
        r.render(want_hit=False)
    st = r.stats()
print(name, st.rays, st.kernel_ms)
