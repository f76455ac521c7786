"""Frames in flight: F contexts with G workgroups each, on F streams, rendering frames round-robin.
Wall time per frame for a whole 4K clutter_500 frame (world 1) and for one rank's bands at world 2/4/8."""
import os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
os.chdir(ROOT)
import torch
import raytracer_amd as rt
from raytracer_amd import dist as rtd

sf = rt.SceneFile("scenes/clutter_500.json", width=3840, height=2160)
desc = sf.desc; desc.use_bvh = 1
W = 3840
configs = [(1, 256), (2, 128), (2, 256), (3, 85), (4, 64), (4, 128), (8, 32)]
if os.environ.get("CONFIGS"):
    configs = [tuple(int(v) for v in c.split("x")) for c in os.environ["CONFIGS"].split(",")]
QUANT = os.environ.get("QUANT") == "1"      # + rt_quantize_device after each frame, like bench.py
CTX = os.environ.get("CTX") == "1"          # make the stream current with torch.cuda.stream, like bench.py
worlds = [int(x) for x in (sys.argv[1:] or ["1", "2", "4", "8"])]
for world in worlds:
    base_ms = None
    for F, G in configs:
        os.environ["RT_WORKGROUPS"] = str(G)
        rs, streams, bufs = [], [], []
        probed = rtd.concurrent_streams(F, torch.device("cuda", 0)) if os.environ.get("PROBE") == "1" and F > 1 else None
        for f in range(F):
            r = rt.Renderer(0)
            r.upload(desc); r.set_camera(sf.camera)
            P = rt.copy_params(sf.params); P.band_rank, P.band_world = 0, world
            r.set_params(P)
            rs.append(r)
            streams.append(probed[f] if probed else torch.cuda.Stream())
            bufs.append(torch.empty(r.local_rows * W * 3, dtype=torch.float64, device="cuda"))
        u8 = [torch.empty(b.numel(), dtype=torch.uint8, device="cuda") for b in bufs]
        for f in range(F):  # calibration frame
            rs[f].render_device(bufs[f].data_ptr(), None, stream=streams[f].cuda_stream)
        torch.cuda.synchronize()
        K = 48
        best = 1e9
        for rep in range(3):
            t0 = time.perf_counter()
            for k in range(K):
                f = k % F
                if CTX:
                    with torch.cuda.stream(streams[f]):
                        st_ = torch.cuda.current_stream().cuda_stream
                        rs[f].render_device(bufs[f].data_ptr(), None, stream=st_)
                        if QUANT: rs[f].quantize_device(bufs[f].data_ptr(), bufs[f].numel(), u8[f].data_ptr(), stream=st_)
                else:
                    rs[f].render_device(bufs[f].data_ptr(), None, stream=streams[f].cuda_stream)
                    if QUANT: rs[f].quantize_device(bufs[f].data_ptr(), bufs[f].numel(), u8[f].data_ptr(), stream=streams[f].cuda_stream)
            torch.cuda.synchronize()
            best = min(best, (time.perf_counter() - t0) / K * 1e3)
        rays = rs[0].stats().rays
        base_ms = base_ms or best
        print(f"world {world}: {F} in flight x {G:3d} workgroups: {best:.3f} ms per frame share, {rays/best/1e3:.0f} Mrays/s "
              f"({base_ms/best:.2f}x)", flush=True)
        for r in rs: r.close()
        del bufs
