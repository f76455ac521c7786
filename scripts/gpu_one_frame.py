"""One frame for a profiler run: scripts/gpu_one_frame.py <scene> <W> <H> [pixel_samples light_samples] -- nothing else
touches the GPU, so every sample / counter belongs to this frame's kernels."""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); os.chdir(ROOT)
import raytracer_amd as rt
name, w, h = sys.argv[1], int(sys.argv[2]), int(sys.argv[3])
sf = rt.SceneFile(f"scenes/{name}.json", width=w, height=h)
P = sf.params
if len(sys.argv) > 5: P.pixel_samples, P.light_samples = int(sys.argv[4]), int(sys.argv[5])
r = rt.Renderer(0)
r.upload(sf.desc); r.set_camera(sf.camera); r.set_params(P)
for _ in range(int(os.environ.get("FRAMES", "2"))):
    r.render(want_hit=False)
st = r.stats()
print(f"{name} {w}x{h}: rays {st.rays} kernel_ms {st.kernel_ms:.3f}")
