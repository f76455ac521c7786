"""N > 1 logic on CPU: two processes over gloo partition one frame the way bench.py does over RCCL
(row bands for Whitted, strata ranges for path tracing) and must reassemble exactly the
one-process frame.  The per-rank renderer here is the oracle (CPU), standing in for rt_render_device:
what is under test is the partition / collective / reassembly code in raytracer_amd/dist.py."""
import os
import sys

import numpy as np
import pytest
import torch
import torch.distributed as dist
import torch.multiprocessing as mp

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _worker(rank, world, port, q):
    sys.path.insert(0, ROOT)
    os.chdir(ROOT)
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    try:
        import raytracer_amd as rt
        from raytracer_amd import dist as rtd
        from tests import oracle_lib, scene_util
        out = {}
        # ---- Whitted: bands -> gather ------------------------------------------------------------
        sf = scene_util.load("mirror_glass_cylinder", 60, 44)  # 44 rows: 6 bands, the last ragged
        P = sf.params
        full, _, _, _ = oracle_lib.render(sf.desc, sf.camera, P, want_hit=False, threads=1)
        g = rtd.BandGatherer(rank, world, P.height, P.width, torch.device("cpu"))
        rows = rtd.band_rows_of(rank, world, P.height)
        mine = torch.from_numpy(np.ascontiguousarray(full[rows]))  # what rt_render_device writes for this rank
        g.mine[: mine.numel()] = mine.reshape(-1)
        frame = g.gather()
        if rank == 0:
            out["bands_equal"] = bool(np.array_equal(frame.numpy(), full))
        # ---- the pipelined form bench.py uses: PPM bytes, two frames in flight ----------------------
        u8_full = (255.999 * full).astype(np.int32).astype(np.uint8)   # rt_quantize_device's conversion
        fp = rtd.FramePipeline(rank, world, P.height, P.width, torch.device("cpu"))
        ok = True
        for k in range(3):  # frame k = the frame with every byte shifted by k (so that a stale buffer shows)
            rgb_t, u8_t = fp.target(k)
            mine8 = torch.from_numpy(np.ascontiguousarray(u8_full[rows] + k))
            u8_t[: mine8.numel()] = mine8.reshape(-1)
            fp.submit(k)
            if k > 0:
                fr = fp.complete(k - 1)
                if rank == 0:
                    ok = ok and bool(np.array_equal(fr.numpy(), u8_full + (k - 1)))
        fr = fp.complete(2)
        if rank == 0:
            out["pipeline_equal"] = ok and bool(np.array_equal(fr.numpy(), u8_full + 2))
        # four frames in flight, the order bench.py issues them in: complete(k - depth) before slot reuse
        fp4 = rtd.FramePipeline(rank, world, P.height, P.width, torch.device("cpu"), depth=4)
        ok4, n_frames = True, 10
        def check4(k):
            fr = fp4.complete(k)
            return rank != 0 or bool(np.array_equal(fr.numpy(), u8_full + k))
        for k in range(n_frames):
            if k >= 4:
                ok4 = check4(k - 4) and ok4
            assert fp4.stream(k) is None
            _, u8_t = fp4.target(k)
            mine8 = torch.from_numpy(np.ascontiguousarray(u8_full[rows] + k))
            u8_t[: mine8.numel()] = mine8.reshape(-1)
            fp4.submit(k)
        for k in range(n_frames - 4, n_frames):
            ok4 = check4(k) and ok4
        if rank == 0:
            out["pipeline4_equal"] = ok4
        # 7 pixels over 2 ranks: slices of 4 pixels, the last one padded; pass k's sums = (rank + 1 + k) * (1 + pixel value index)
        ap = rtd.AccumulatorPipeline(rank, world, 7, torch.device("cpu"))
        ramp = torch.arange(1, 22, dtype=torch.float64)
        def expect(k, scale=1.0):
            return (ramp * float(sum(r + 1 + k for r in range(world))) * scale).numpy()
        ok = True
        for k in range(3):
            t = ap.target(k)
            assert t.numel() == 24
            t[:21] = ramp * float(rank + 1 + k)
            ap.submit(k)
            if k > 0:
                got = ap.complete(k - 1)
                if rank == 0:
                    ok = ok and got.shape == (21,) and bool(np.array_equal(got.numpy(), expect(k - 1)))
        # the last pass with a finish step of another dtype (the non-linear stages + byte conversion of bench.py)
        got = ap.complete(2, finish=lambda sums, n_px: (sums * 0.5).to(torch.float32))
        if rank == 0:
            out["accum_pipeline"] = ok and got.dtype == torch.float32 and bool(np.array_equal(got.numpy(), expect(2, 0.5).astype(np.float32)))
        # ---- path trace: strata ranges -> reduce -> resolve --------------------------------------
        pt = scene_util.load("pathtrace", 24, 16)
        Q = pt.params
        Q.pixel_samples, Q.light_samples = 9, 2
        ref_rgb, _, _, ref_acc = oracle_lib.render(pt.desc, pt.camera, Q, want_hit=False, want_accum=True, threads=1)
        lo, hi = rtd.strata_range(rank, world, 9)
        R = rt.copy_params(Q)
        R.sample_begin, R.sample_end = lo, hi
        _, _, _, acc = oracle_lib.render(pt.desc, pt.camera, R, want_hit=False, want_accum=True, threads=1)
        t = torch.from_numpy(acc.copy())
        rtd.reduce_accumulators(t, world)
        if rank == 0:
            out["accum_close"] = bool(np.allclose(t.numpy(), ref_acc, rtol=1e-13, atol=1e-15))
            c = t.numpy() / Q.pixel_samples          # RayTracer.cpp:301-315 with tone map NONE, exposure 1
            c = np.clip(np.power(c, 1.0 / 2.2), 0.0, 1.0)
            out["resolved_close"] = bool(np.abs(c - ref_rgb).max() < 1e-12)
        # an empty share (more ranks than strata) must contribute exactly zero
        lo, hi = rtd.strata_range(rank, 4, 2)
        out_range = (lo, hi)
        if rank == 0:
            out["ranges"] = [rtd.strata_range(r, 4, 2) for r in range(4)]
            q.put(out)
    finally:
        dist.destroy_process_group()


def test_two_ranks_reassemble_the_frame():
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    port = 29500 + (os.getpid() % 2000)
    procs = [ctx.Process(target=_worker, args=(r, 2, port, q)) for r in range(2)]
    for p in procs:
        p.start()
    out = q.get(timeout=240)
    for p in procs:
        p.join(timeout=60)
        assert p.exitcode == 0
    assert out["bands_equal"] and out["accum_close"] and out["resolved_close"]
    assert out["pipeline_equal"] and out["pipeline4_equal"] and out["accum_pipeline"]
    assert out["ranges"] == [(2, 2), (0, 1), (2, 2), (1, 2)]


def test_band_partition_covers_every_row_once():
    from raytracer_amd import dist as rtd
    for H in (1, 7, 8, 9, 800, 2160):
        for world in (1, 2, 3, 8):
            rows = sorted(j for r in range(world) for j in rtd.band_rows_of(r, world, H))
            assert rows == list(range(H))
            assert all(len(rtd.band_rows_of(r, world, H)) <= rtd.max_band_rows(world, H) for r in range(world))
