#!/usr/bin/env python3
"""bench.py -- headline benchmark of the MI355X render loop (driver contract in the task brief).

    python bench.py [--gpus N] [--steps K] [--warmup W] [--workload clutter_500_4k|pathtrace_1080p]

A "step" is one pass of the hot path over one frame of the workload:

  clutter_500_4k   (default; BASELINE.json configs[2]) scenes/clutter_500.json at 3840x2160,
                   Whitted ("phong") mode, nbounces 8, BVH tie rule (the reference renders the
                   same bytes with "bvh": false as shipped).  Metric: Mrays/s,
                   ray = one closest-hit query = one Scene::intersect call of the reference.
  pathtrace_1080p  (configs[3]) scenes/pathtrace.json at 1920x1080, lightSamples 16, with
                   --spp strata per step (default 16; the 1024-spp frame is 64 such steps --
                   samples are independent, so spp/s does not depend on the count).

With N > 1 (launched by torch.distributed.run, one rank per GPU) ONE frame is split over the
ranks -- row bands for Whitted, strata ranges for path tracing -- and recombined on rank 0's
device through RCCL inside the timed region: a gather of the bands as PPM bytes (the reference's
int(255.999 * c), done on the device: an eighth of the fp64 data) / a reduce of the fp64 partial
sums.  Frames are pipelined two deep, so the collective of frame k overlaps the kernels of frame
k+1.  Total work is fixed, so "scaling" is "strong".

Rank 0 prints ONE JSON line.  `roofline` prices the persistent render kernel against the fp64
VALU peak WITHOUT FMA credit (contraction is off for parity): algorithmic ops = 24 per box test
+ 33 per sphere + 62 per triangle + 150 per cylinder test, counted in the
reference's traversal order -- which is exactly the order this kernel walks.  `cpu_baseline` is
the pinned fp64 CPU restatement (oracle/) on the same frame on this box's host cores.
"""
import argparse
import ctypes as C
import json
import os
import sys
import time

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)
os.chdir(ROOT)

FP64_VALU_PEAK_TOPS = 78.6 / 2.0  # MI355X fp64 vector 78.6 TFLOP/s counts an FMA as 2; no FMA credit here
OPS_BOX, OPS_SPHERE, OPS_TRI, OPS_CYL = 24, 33, 62, 150


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=20)
    ap.add_argument("--warmup", type=int, default=3)
    ap.add_argument("--workload", default="clutter_500_4k", choices=["clutter_500_4k", "pathtrace_1080p"])
    ap.add_argument("--spp", type=int, default=16, help="path trace: strata per step")
    ap.add_argument("--in-flight", type=int, default=0,
                    help="Whitted: frames in flight, one render context + stream each (default 4)")
    ap.add_argument("--workgroups", type=int, default=-1,
                    help="workgroups per context (default: 0 = fill the chip when 1 frame is in flight, else 64)")
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--emulate-world", type=int, default=0,
                    help="rehearsal on one GPU: render rank 0's share of an N-rank job (no collective); the printed "
                         "value then refers to that share only")
    args = ap.parse_args()

    import numpy as np
    import torch
    import torch.distributed as dist

    import raytracer_amd as rt
    from raytracer_amd import dist as rtd

    world = int(os.environ.get("WORLD_SIZE", "1"))
    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    if world > 1:
        torch.cuda.set_device(local_rank)
        dist.init_process_group("nccl", device_id=torch.device("cuda", local_rank))
    dev = torch.device("cuda", local_rank)
    torch.cuda.set_device(dev)

    pathtrace = args.workload == "pathtrace_1080p"
    if pathtrace:
        sf = rt.SceneFile("scenes/pathtrace.json", width=1920, height=1080)
        base = sf.params
        side = int(np.sqrt(args.spp))
        base.pixel_samples, base.light_samples = side * side, 16
        desc = sf.desc
    else:
        sf = rt.SceneFile("scenes/clutter_500.json", width=3840, height=2160)
        base = sf.params
        desc = sf.desc
        desc.use_bvh = 1
    W, H = base.width, base.height

    # Reference-order work of one frame: the roofline's numerator is counted on the
    # reference's own traversal, so one untimed frame is rendered with the visit-for-visit walk
    # (RT_TRAVERSAL=walk) on this rank's share, before the production context is created.
    os.environ["RT_TRAVERSAL"] = "walk"
    rw = rt.Renderer(local_rank)
    os.environ.pop("RT_TRAVERSAL")
    P = rt.copy_params(base)
    if pathtrace:
        P.sample_begin, P.sample_end = rtd.strata_range(rank, world, base.pixel_samples)
    else:
        P.band_rank, P.band_world = (0, args.emulate_world) if (args.emulate_world > 0 and world == 1) else (rank, world)
    rw.upload(desc)
    rw.set_camera(sf.camera)
    rw.set_params(P)
    rw.render(want_hit=False)
    ref_st = rw.stats()
    ref_counts = [ref_st.rays, ref_st.box_tests, ref_st.sphere_tests, ref_st.triangle_tests, ref_st.cylinder_tests]
    rw.close()

    # Frames in flight.  Whitted: F render contexts of G workgroups each, one stream per context, frames
    # dealt round-robin -- the frames share the chip's 256 CUs, which hides the per-level latency of a
    # frame's thin levels (and of a rank's 1/N share of a frame); the gather of frame k (RCCL's stream)
    # overlaps the kernels of the frames after it.  Whitted frames travel as PPM bytes
    # (rt_quantize_device), path-trace passes as fp64 partial sums (the division / tone map / gamma come
    # after the reduction); path tracing keeps one context and two buffers.
    F = 1 if pathtrace else (args.in_flight if args.in_flight > 0 else 4)
    streams = None
    while F > 1 and streams is None:
        try:
            streams = rtd.concurrent_streams(F, dev)  # streams that do not share a hardware queue (measured)
        except Exception:  # fewer streams overlap than asked for (or the probe is unavailable): fewer frames in flight
            F -= 1
    G = args.workgroups if args.workgroups >= 0 else (0 if F == 1 else 64)
    rs = []
    for _ in range(F):
        r = rt.Renderer(local_rank)
        r.upload(desc)
        r.set_camera(sf.camera)
        r.set_params(P)
        r.set_workgroups(G)
        rs.append(r)
    rows = rs[0].local_rows
    depth = max(F, 2)  # slots; a slot is free again once its frame was completed
    if pathtrace:
        pipe = rtd.AccumulatorPipeline(world, H * W * 3, dev, depth=depth)
    else:
        pipe = rtd.FramePipeline(rank, world, H, W, dev, band_rows=P.band_rows, depth=depth, streams=streams)

    import contextlib

    def on_stream(k):
        s = None if pathtrace else pipe.stream(k)
        return torch.cuda.stream(s) if s is not None else contextlib.nullcontext()

    def enqueue(k, ev=None):
        r = rs[k % F]
        with on_stream(k):
            stream = torch.cuda.current_stream().cuda_stream
            if pathtrace:
                accum = pipe.target(k)
                if ev: ev[0].record()
                r.accumulate_device(accum.data_ptr(), stream=stream)
                if ev: ev[1].record()
            else:
                _, u8 = pipe.target(k)
                if ev: ev[0].record()
                r.render_device_u8(u8.data_ptr(), None, None, stream=stream)  # PPM bytes straight from the kernels
                if ev: ev[1].record()
            pipe.submit(k)

    def finish(k):
        with on_stream(k):
            out = pipe.complete(k)
            if pathtrace and rank == 0:
                rs[0].resolve_device(out.data_ptr(), H * W, stream=torch.cuda.current_stream().cuda_stream)

    def run(n, events=None):
        for k in range(n):
            enqueue(k, events[k] if events else None)
            if k >= depth - 1:
                finish(k - (depth - 1))  # frees the slot frame k + 1 renders into
        for k in range(max(n - (depth - 1), 0), n):
            finish(k)

    def sync_all():
        torch.cuda.synchronize()
        if world > 1:
            dist.barrier()
            torch.cuda.synchronize()

    run(F)  # every context's first frame sizes its node store (synchronous): not a warm-up step
    sync_all()
    run(args.warmup)
    sync_all()
    # per-launch duration of the render kernels: HIP events on the stream they are launched on
    events = [(torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)) for _ in range(args.steps)]
    t0 = time.perf_counter()
    run(args.steps, events)
    sync_all()
    elapsed = time.perf_counter() - t0
    kernel_ms = [a.elapsed_time(b) for a, b in events]
    st = rs[(args.steps - 1) % F].stats()

    t = torch.tensor([elapsed], dtype=torch.float64, device=dev)
    assert st.rays == ref_counts[0], "the production traversal must trace exactly the reference's rays"
    cnt = torch.tensor(ref_counts, dtype=torch.float64, device=dev)
    done = torch.tensor([st.box_tests, st.prim_tests], dtype=torch.float64, device=dev)
    kms = torch.tensor([float(np.mean(kernel_ms))], dtype=torch.float64, device=dev)
    if world > 1:
        dist.all_reduce(t, op=dist.ReduceOp.MAX)
        dist.all_reduce(cnt, op=dist.ReduceOp.SUM)
        dist.all_reduce(done, op=dist.ReduceOp.SUM)
        dist.all_reduce(kms, op=dist.ReduceOp.MAX)
    elapsed = float(t.item())
    rays, boxes, sph, tri, cyl = (float(x) for x in cnt.tolist())
    k_ms = float(kms.item())

    if rank == 0:
        ms_per_step = elapsed / args.steps * 1e3
        mrays = rays * args.steps / elapsed / 1e6
        alg_ops = OPS_BOX * boxes + OPS_SPHERE * sph + OPS_TRI * tri + OPS_CYL * cyl  # per frame, all ranks
        # per GPU: ops of one launch / its duration.  With F launches in flight on F streams a launch has a
        # share of the chip only, so the chip's rate is the ops of all launches of the timed region over
        # that region (a lower bound: it includes the gaps between kernels).
        achieved = alg_ops / world / ((k_ms if F == 1 else ms_per_step) * 1e-3) / 1e12
        line = {
            "metric": "Mrays/s" if not pathtrace else "spp/s",
            "value": mrays if not pathtrace else (base.pixel_samples / (elapsed / args.steps)),
            "unit": "Mrays/s" if not pathtrace else "spp/s at 1920x1080",
            "n_gpus": world, "steps": args.steps, "warmup": args.warmup,
            "ms_per_step": ms_per_step, "higher_is_better": True, "scaling": "strong",
            "vs_baseline": None, "dtype": "f64", "data": "reference scene file (scenes/%s.json), resized" %
                                                         ("pathtrace" if pathtrace else "clutter_500"),
            "config": {"workload": ("pathtrace.json 1920x1080, %d strata/step, 16 light samples" % base.pixel_samples)
                       if pathtrace else "clutter_500.json 3840x2160 Whitted nbounces=8, BVH tie rule",
                       "rays_per_step": rays, "mrays_per_s": mrays,
                       "parallelism": ("strata-range x%d + RCCL reduce" if pathtrace else "row-band x%d + RCCL gather") % world},
            "roofline": {"bound": "valu_fp64", "achieved": achieved, "peak": FP64_VALU_PEAK_TOPS, "unit": "TFLOP/s",
                         "frac": achieved / FP64_VALU_PEAK_TOPS, "traffic": None,
                         "kernel": ("bw_kernel + bw_accumulate_path_kernel per batch of strata (per depth: trace, shadow, shade; combine)" if pathtrace
                                    else "fw_roots_kernel (level 0) + fw_kernel (depth >= 1: fused trace/shadow/shade per level, combine, tone map)"),
                         "kernel_ms": k_ms, "launches_in_flight": F, "workgroups_per_launch": G or "all",
                         # one launch's own rate (ops of one launch / its duration): with F launches in flight it
                         # runs on 1/F of the chip, so `achieved` above is the ops of all launches over the timed
                         # region instead; with --in-flight 1 the two definitions coincide
                         "achieved_per_launch": alg_ops / world / (k_ms * 1e-3) / 1e12,
                         "ops_per_ray": alg_ops / max(rays, 1.0),
                         "reference_box_tests_per_ray": boxes / max(rays, 1.0),
                         # counters of the production traversal are compiled out (-DRT_FBVH_COUNTS=1 turns them on;
                         # clutter_500: 18.7 fp32 node tests, 1.08 gate and 1.08 primitive tests per ray)
                         "executed_box_tests_per_ray": (float(done[0].item()) / max(rays, 1.0)) or None,
                         "executed_prim_tests_per_ray": (float(done[1].item()) / max(rays, 1.0)) or None,
                         "note": "fp64 ops without FMA credit (contraction off for parity); peak = 78.6/2"},
        }
        # HBM bytes per frame from the PMC passes (collected offline with rocprofv3 --pmc, see the file): not
        # measurable inside this process, so it is quoted from the committed profile of the same workload
        tpath = os.path.join(ROOT, "profiles", "r01_i_traffic.json")
        if not pathtrace and world == 1 and os.path.exists(tpath):
            tr = json.load(open(tpath))
            line["roofline"]["traffic"] = tr["fetch_bytes_raw"] + tr["write_bytes"]
            line["roofline"]["traffic_source"] = tr["source"]
        if not args.no_cpu_baseline and world == 1:
            line["cpu_baseline"] = cpu_baseline(sf, desc, base, pathtrace)
        print(json.dumps(line), flush=True)
    if world > 1:
        dist.barrier()
        dist.destroy_process_group()


def cpu_baseline(sf, desc, base, pathtrace):
    """The pinned oracle (oracle/liboracle.so, -O2 -fopenmp) on this box's host cores: checker code
    used here only as the reported baseline."""
    import raytracer_amd as rt
    from tests import oracle_lib
    threads = min(oracle_lib.load().oracle_max_threads(), os.cpu_count() or 1)
    P = rt.copy_params(base)
    if pathtrace:
        P.width, P.height = 480, 270  # bounded sample: 1/16 of the frame's pixels, same strata
        sf.resize(480, 270)
        sample = "480x270 crop-equivalent (1/16 of the pixels), all strata of one step"
    else:
        sample = "the full 3840x2160 frame, rendered repeatedly for >= 10 s"
    t0 = time.perf_counter()
    n = 0
    rays = 0
    while True:
        _, _, cnt, _ = oracle_lib.render(desc, sf.camera, P, want_hit=False, threads=threads)
        rays += cnt["rays"]
        n += 1
        if time.perf_counter() - t0 > 10.0 or n >= 50:
            break
    dt = time.perf_counter() - t0
    if pathtrace:
        value = P.pixel_samples * n / dt / 16.0  # spp/s referred to the full 1080p frame
        unit = "spp/s at 1920x1080"
    else:
        value = rays / dt / 1e6
        unit = "Mrays/s"
    return {"value": value, "unit": unit, "cores": threads, "kind": "port",
            "sample": sample + f" ({n} frames, {dt:.1f} s)",
            "note": "our fp64 CPU restatement, byte-identical to the reference; the reference itself, as built "
                    "by its Makefile, measured 0.6-2.1 Mrays/s on 8 cores (BASELINE.md)"}


if __name__ == "__main__":
    main()
