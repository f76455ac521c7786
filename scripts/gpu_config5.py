"""BASELINE config 5's workload on one GPU: final_textured_lens.json (final.json + textures + thin lens) at 3840x2160,
pixelSamples 4096 (64x64 strata), Uncharted2 -- a range of its strata is rendered and timed:
   scripts/gpu_config5.py [n_strata]   (default 64 of the 4096)"""
import os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); os.chdir(ROOT)
import numpy as np
import raytracer_amd as rt
n = int(sys.argv[1]) if len(sys.argv) > 1 else 64
sf = rt.SceneFile("scenes/final_textured_lens.json", tonemap="uncharted2", width=3840, height=2160)
P = sf.params
P.pixel_samples = 4096
ls = P.light_samples
P.sample_begin, P.sample_end = 0, n
r = rt.Renderer(0)
r.load(sf, P)
rgb, _ = r.render(want_hit=False)   # sizes the node store
ms = []
for _ in range(2):
    r.render(want_hit=False); ms.append(r.stats().kernel_ms)
st = r.stats()
m = float(np.median(ms))
print(f"final_textured_lens 3840x2160, strata 0..{n-1} of 4096, {ls} light samples: {m:.1f} ms, {st.rays} rays = {st.rays/m/1e6:.1f} Grays/s, "
      f"{n/m*1e3:.1f} spp/s at 4K; finite {bool(np.isfinite(rgb).all())}, 4096 spp would take {4096/n*m/1e3:.1f} s")
