"""Per-rank kernel time of a band-sharded 4K clutter_500 frame (what one rank of an N-GPU job renders)."""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); os.chdir(ROOT)
import numpy as np
import raytracer_amd as rt
sf = rt.SceneFile("scenes/clutter_500.json", width=3840, height=2160)
desc = sf.desc; desc.use_bvh = 1
r = rt.Renderer(0)
r.upload(desc); r.set_camera(sf.camera)
base = None
for world in (1, 2, 4, 8):
    worst = 0.0
    for rank in range(world):
        P = rt.copy_params(sf.params); P.band_rank, P.band_world = rank, world
        r.set_params(P)
        r.render(want_hit=False)
        ms = []
        for _ in range(4):
            r.render(want_hit=False); ms.append(r.stats().kernel_ms)
        worst = max(worst, float(np.median(ms)))
    base = base or worst
    print(f"world {world}: slowest rank {worst:.3f} ms  -> kernel-only strong-scaling efficiency {base / (world * worst):.2f}", flush=True)
