#!/usr/bin/env python3
"""bench.py -- headline benchmark of the MI355X render loop (driver contract in the task brief).

    python bench.py [--gpus N] [--steps K] [--warmup W] [--workload both|clutter_500_4k|pathtrace_1080p]

BASELINE.json's metric has two halves, and the default invocation (`--workload both`) measures both
and prints them on ONE JSON line:

  clutter_500_4k   (configs[2]; the line's `value`) scenes/clutter_500.json at 3840x2160, Whitted
                   ("phong") mode, nbounces 8, BVH tie rule (the reference renders the same bytes with
                   "bvh": false as shipped).  Metric: Mrays/s, ray = one closest-hit
                   query = one Scene::intersect call of the reference.  A step = one frame; K steps are
                   timed --repeat times (default 10) and the MEDIAN region gives `value`.
  pathtrace_1080p  (configs[3]; the line's `pathtrace` block) scenes/pathtrace.json at 1920x1080,
                   pixelSamples 1024 = 32x32 strata (JSON "lightsample": 1024, RayTracer.cpp:138-139),
                   lightSamples 16 (the shipped default).  Metric: spp/s = pixel-samples per second /
                   pixel count, plus Mrays/s.  A step = one whole 1024-spp frame (--pt-steps, default 2).

With N > 1 (launched by torch.distributed.run, one rank per GPU) ONE frame is split over the ranks --
row bands for Whitted, strata ranges for path tracing -- and recombined on rank 0's device through RCCL
inside the timed region: a gather of the bands as PPM bytes / a reduce-scatter of the fp64 partial sums
before the non-linear stages, which each rank applies to its slice, and a gather of the finished slices
(SURVEY.md 8(e)).  Total work is fixed, so "scaling" is "strong".  `python bench.py --gpus N` without
WORLD_SIZE in the environment starts the N ranks itself (fresh children, before this process touches a
GPU) and fails loudly when fewer than N GPUs are visible; `--mode group` times the one-process form
(rt_group_render_u8 over devices 0..N-1) instead.

`roofline` prices the render kernels against the fp64 VALU peak WITHOUT FMA credit (contraction is off
for parity): algorithmic ops = 24 per box test + 33 per sphere + 62 per triangle + 150 per cylinder test
(SURVEY.md 8(d)), counted on the REFERENCE's traversal order by one untimed frame with
RT_TRAVERSAL=walk, divided by the kernels' duration (HIP events on the stream they run on).  Beside
`frac`: `frac_walked` (the same numerator without the tests of rays that are counted but never walked:
shadow rays of refracting Whitted hits, of dark and clear path-traced hits), for the path tracer
`frac_with_sampling` (plus a lower bound of the reference's per-light-sample arithmetic), and what the
production traversal really executes (`executed_*`, from one untimed frame of the -DRT_FBVH_COUNTS=1
build in a child process).  `cpu_baseline` is the pinned fp64 CPU restatement
(oracle/) on a bounded sample of the same workload on this box's host cores.
"""
import argparse
import contextlib
import json
import os
import subprocess
import sys
import time

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)
os.chdir(ROOT)

FP64_VALU_PEAK_TOPS = 78.6 / 2.0  # MI355X fp64 vector 78.6 TFLOP/s counts an FMA as 2; no FMA credit here
OPS_BOX, OPS_SPHERE, OPS_TRI, OPS_CYL = 24, 33, 62, 150
COUNTS_LIB = os.path.join(ROOT, "raytracer_amd", "csrc", "variants", "libraytracer_amd_counts.so")
PT_SPP, PT_LIGHT_SAMPLES = 1024, 16


def scene_setup(workload, spp=PT_SPP):
    import raytracer_amd as rt
    if workload == "pathtrace_1080p":
        sf = rt.SceneFile("scenes/pathtrace.json", width=1920, height=1080)
        base = sf.params
        base.pixel_samples, base.light_samples = spp, PT_LIGHT_SAMPLES
        desc = sf.desc
    else:
        sf = rt.SceneFile("scenes/clutter_500.json", width=3840, height=2160)
        base = sf.params
        desc = sf.desc
        desc.use_bvh = 1
    return sf, desc, base


def executed_counts_child(workload, spp):
    """Child process: one frame with the counting build; prints the production traversal's test counts."""
    import raytracer_amd as rt
    sf, desc, base = scene_setup(workload, spp)
    if workload == "pathtrace_1080p":
        base.sample_begin, base.sample_end = 0, min(16, spp)  # 16 strata are plenty for per-ray averages
    r = rt.Renderer(0)
    r.upload(desc); r.set_camera(sf.camera); r.set_params(base)
    r.render_u8() if workload != "pathtrace_1080p" else r.render(want_hit=False)
    st = r.stats()
    print(json.dumps({"rays": st.rays, "box": st.box_tests, "sphere": st.sphere_tests, "tri": st.triangle_tests,
                      "cyl": st.cylinder_tests, "walks": st.rays_traversed}))


def executed_counts(workload, spp):
    """Test counts the production (filter-BVH) traversal executes: its counters are compiled out of the
    production build, so one untimed frame is rendered by the -DRT_FBVH_COUNTS=1 build in a child process
    (started before this process touches the GPU).  None when that build is absent."""
    if not os.path.exists(COUNTS_LIB):
        return None
    env = dict(os.environ, RT_AMD_LIB=COUNTS_LIB)
    env.pop("RT_TRAVERSAL", None)
    try:
        out = subprocess.run([sys.executable, os.path.abspath(__file__), "--counts-child", workload, "--spp", str(spp)],
                             env=env, capture_output=True, text=True, timeout=240, check=True).stdout
        c = json.loads(out.strip().splitlines()[-1])
    except Exception as e:  # the line is still valid without it
        print(f"[bench] executed-count child failed: {e}", file=sys.stderr)
        return None
    rays = max(c["rays"], 1)
    prim = c["sphere"] + c["tri"] + c["cyl"]
    return {"rays_traversed_per_ray": c.get("walks", 0) / rays,
            "rays_traversed_note": "every ray of the reference (one Scene::intersect call) is counted; a ray whose answer cannot "
                                   "reach the image is not walked: shadow rays of a Whitted hit that refracts (RayTracer.cpp:798 "
                                   "replaces its local colour), of a path-traced hit that faces away from the whole light or "
                                   "whose shaft to the light is empty",
            "executed_box_tests_per_ray": c["box"] / rays, "executed_prim_tests_per_ray": prim / rays,
            "executed_ops_per_ray": (OPS_BOX * c["box"] + OPS_SPHERE * c["sphere"] + OPS_TRI * c["tri"] + OPS_CYL * c["cyl"]) / rays,
            "executed_note": "filter-BVH traversal: box tests = fp32 node tests (21 instructions each, priced like an fp64 box "
                             "test here) + fp64 gate tests; from one untimed frame of the -DRT_FBVH_COUNTS=1 build"}


def add_executed_rates(line, roof, rays_per_step, ms_per_step):
    """The conservative companions of `value` and `roofline.achieved`: only the rays whose traversal was started, and
    only the box / primitive tests the production traversal executed (priced like the reference's)."""
    per_s = rays_per_step / (ms_per_step * 1e-3)
    line["mrays_per_s_traversed_only"] = per_s * roof["rays_traversed_per_ray"] / 1e6
    roof["achieved_executed"] = roof["executed_ops_per_ray"] * per_s / line["n_gpus"] / 1e12
    roof["frac_executed"] = roof["achieved_executed"] / FP64_VALU_PEAK_TOPS


def quoted_profile(name):
    """Numbers that cannot be measured inside this process (rocprofv3 PMC passes), quoted from the committed
    profile with its source; None if the file is absent."""
    p = os.path.join(ROOT, "profiles", name)
    if not os.path.exists(p):
        return None
    d = json.load(open(p))
    d["quoted_from"] = "profiles/" + name
    return d


class Bench:
    def __init__(self, args):
        import numpy as np
        import torch
        import torch.distributed as dist
        import raytracer_amd as rt
        from raytracer_amd import dist as rtd
        self.np, self.torch, self.dist, self.rt, self.rtd, self.args = np, torch, dist, rt, rtd, args
        self.world = int(os.environ.get("WORLD_SIZE", "1"))
        self.rank = int(os.environ.get("RANK", "0"))
        self.local_rank = int(os.environ.get("LOCAL_RANK", "0"))
        if self.world > 1:
            torch.cuda.set_device(self.local_rank)
            dist.init_process_group("nccl", device_id=torch.device("cuda", self.local_rank))
        self.dev = torch.device("cuda", self.local_rank)
        torch.cuda.set_device(self.dev)

    def sync_all(self):
        self.torch.cuda.synchronize()
        if self.world > 1:
            self.dist.barrier()
            self.torch.cuda.synchronize()

    def reference_counts(self, sf, desc, P):
        """Reference-order work of this rank's share of one frame: one untimed frame with the
        visit-for-visit walk of the reference tree (RT_TRAVERSAL=walk, read by rt_create)."""
        os.environ["RT_TRAVERSAL"] = "walk"
        rw = self.rt.Renderer(self.local_rank)
        os.environ.pop("RT_TRAVERSAL")
        rw.upload(desc); rw.set_camera(sf.camera); rw.set_params(P)
        if P.mode == self.rt.RT_MODE_PATHTRACE:
            t = self.torch.empty(P.width * rw.local_rows * 3, dtype=self.torch.float64, device=self.dev)
            rw.accumulate_device(t.data_ptr(), stream=self.torch.cuda.current_stream().cuda_stream)
        else:
            rw.render_u8()
        st = rw.stats()
        rw.close()
        return [st.rays, st.box_tests, st.sphere_tests, st.triangle_tests, st.cylinder_tests, st.unwalked_ops]

    def reduce_counts(self, ref_counts, elapsed_list, kernel_ms):
        torch, dist = self.torch, self.dist
        t = torch.tensor(elapsed_list, dtype=torch.float64, device=self.dev)
        cnt = torch.tensor(ref_counts, dtype=torch.float64, device=self.dev)
        kms = torch.tensor([kernel_ms], dtype=torch.float64, device=self.dev)
        if self.world > 1:
            dist.all_reduce(t, op=dist.ReduceOp.MAX)
            dist.all_reduce(cnt, op=dist.ReduceOp.SUM)
            dist.all_reduce(kms, op=dist.ReduceOp.MAX)
        return t.tolist(), cnt.tolist(), float(kms.item())

    # ---- Whitted: clutter_500 at 4K ------------------------------------------------------------------
    def whitted(self):
        args, rt, rtd, torch, np = self.args, self.rt, self.rtd, self.torch, self.np
        sf, desc, base = scene_setup("clutter_500_4k")
        W, H = base.width, base.height
        P = rt.copy_params(base)
        P.band_rank, P.band_world = (0, args.emulate_world) if (args.emulate_world > 0 and self.world == 1) else (self.rank, self.world)
        ref_counts = self.reference_counts(sf, desc, P)

        # Frames in flight: F render contexts of G workgroups each, one stream per context, frames dealt
        # round-robin -- the frames share the chip's 256 CUs, which hides the per-level latency of a frame's
        # thin levels (and of a rank's 1/N share of a frame); the gather of frame k (RCCL's stream) overlaps
        # the kernels of the frames after it.  Frames travel as PPM bytes written by the kernels' pixel store.
        F = args.in_flight if args.in_flight > 0 else 4
        streams = None
        while F > 1 and streams is None:
            try:
                streams = rtd.concurrent_streams(F, self.dev)  # streams that do not share a hardware queue (measured)
            except Exception:  # fewer streams overlap than asked for (or the probe is unavailable)
                F -= 1
        G = args.workgroups if args.workgroups >= 0 else (0 if F == 1 else 64)
        rs = []
        for _ in range(F):
            r = rt.Renderer(self.local_rank)
            r.upload(desc); r.set_camera(sf.camera); r.set_params(P)
            r.set_workgroups(G)
            rs.append(r)
        depth = max(F, 2)  # slots; a slot is free again once its frame was completed
        pipe = rtd.FramePipeline(self.rank, self.world, H, W, self.dev, band_rows=P.band_rows, depth=depth, streams=streams)

        def on_stream(k):
            s = pipe.stream(k)
            return torch.cuda.stream(s) if s is not None else contextlib.nullcontext()

        def enqueue(k, ev=None):
            with on_stream(k):
                _, u8 = pipe.target(k)
                if ev: ev[0].record()
                rs[k % F].render_device_u8(u8.data_ptr(), None, None, stream=torch.cuda.current_stream().cuda_stream)
                if ev: ev[1].record()
                pipe.submit(k)

        def finish(k):
            with on_stream(k):
                pipe.complete(k)

        def run(n, events=None):
            for k in range(n):
                enqueue(k, events[k] if events else None)
                if k >= depth - 1:
                    finish(k - (depth - 1))  # frees the slot frame k + 1 renders into
            for k in range(max(n - (depth - 1), 0), n):
                finish(k)

        run(F)  # every context's first frame sizes its node store (synchronous): not a warm-up step
        self.sync_all()
        run(args.warmup)
        self.sync_all()
        # EXACTLY K steps per timed region, bracketed by barrier + synchronize; the region is repeated and the
        # median region is the headline (one 30 ms window is a thin sample); every region is listed
        regions, kernel_ms = [], []
        for _ in range(max(args.repeat, 1)):
            events = [(torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)) for _ in range(args.steps)]
            t0 = time.perf_counter()
            run(args.steps, events)
            self.sync_all()
            regions.append(time.perf_counter() - t0)
            kernel_ms.append(float(np.mean([a.elapsed_time(b) for a, b in events])))
        st = rs[(args.steps - 1) % F].stats()
        assert st.rays == ref_counts[0], "the production traversal must trace exactly the reference's rays"

        # one frame alone on the whole chip: the latency a caller that renders one frame at a time sees
        latency_ms = None
        if self.world == 1 and args.emulate_world == 0:
            r1 = rt.Renderer(self.local_rank)
            r1.upload(desc); r1.set_camera(sf.camera); r1.set_params(P)
            u8 = torch.zeros(H * W * 3, dtype=torch.uint8, device=self.dev)
            lat = []
            for k in range(12):
                r1.render_device_u8(u8.data_ptr(), None, None, stream=torch.cuda.current_stream().cuda_stream)
                torch.cuda.synchronize()
                if k >= 2:
                    lat.append(r1.stats().kernel_ms)
            latency_ms = float(np.median(lat))
            r1.close()
        for r in rs:
            r.close()

        regions, cnt, k_ms = self.reduce_counts(ref_counts, regions, float(np.median(kernel_ms)))
        rays, boxes, sph, tri, cyl, unwalked_ops = cnt
        elapsed = float(np.median(regions))
        ms_per_step = elapsed / args.steps * 1e3
        mrays = rays * args.steps / elapsed / 1e6
        alg_ops = OPS_BOX * boxes + OPS_SPHERE * sph + OPS_TRI * tri + OPS_CYL * cyl  # per frame, all ranks
        # per GPU: ops of one launch / its duration.  With F launches in flight on F streams a launch has a
        # share of the chip only, so the chip's rate is the ops of all launches of the timed region over
        # that region (a lower bound: it includes the gaps between kernels).
        achieved = alg_ops / self.world / ((k_ms if F == 1 else ms_per_step) * 1e-3) / 1e12
        walked = (alg_ops - unwalked_ops) / alg_ops
        roof = {"bound": "valu_fp64", "achieved": achieved, "peak": FP64_VALU_PEAK_TOPS, "unit": "TFLOP/s",
                "frac": achieved / FP64_VALU_PEAK_TOPS, "traffic": None,
                # the same numerator without the reference-order tests of rays that are counted but not walked (shadow rays of
                # hits whose local colour RayTracer.cpp:798 overwrites), counted by the same walk frame
                "frac_walked": achieved * walked / FP64_VALU_PEAK_TOPS, "walked_ops_share": walked,
                "kernel": "fw_roots_kernel (level 0: trace, shadow rays, shade) + fw_kernel (depth >= 1, per level: trace + child rays, then the "
                          "lights of the hits whose local colour is needed; combine, tone map)",
                "kernel_ms": k_ms, "launches_in_flight": F, "workgroups_per_launch": G or "all",
                # one launch's own rate (ops of one launch / its duration): with F launches in flight it runs on
                # 1/F of the chip, so `achieved` is the ops of all launches over the timed region instead
                "achieved_per_launch": alg_ops / self.world / (k_ms * 1e-3) / 1e12,
                "latency_ms_single_frame": latency_ms,
                "achieved_single_frame": (alg_ops / (latency_ms * 1e-3) / 1e12) if latency_ms else None,
                "ops_per_ray": alg_ops / max(rays, 1.0),
                "reference_box_tests_per_ray": boxes / max(rays, 1.0),
                "note": "fp64 ops without FMA credit (contraction off for parity); peak = 78.6/2; numerator counted on the "
                        "reference's traversal order (SURVEY 8(d)), executed_* is what the filter-BVH traversal really does"}
        line = {"metric": "Mrays/s", "value": mrays, "unit": "Mrays/s", "n_gpus": self.world, "steps": args.steps,
                "warmup": args.warmup, "ms_per_step": ms_per_step, "higher_is_better": True, "scaling": "strong",
                "vs_baseline": None, "dtype": "f64", "data": "reference scene file (scenes/clutter_500.json), resized",
                "config": {"workload": "clutter_500.json 3840x2160 Whitted nbounces=8, BVH tie rule",
                           "rays_per_step": rays, "mrays_per_s": mrays, "parallelism": "row-band x%d + RCCL gather" % self.world,
                           "timed_regions": len(regions), "ms_per_step_median": ms_per_step,
                           "ms_per_step_min": min(regions) / args.steps * 1e3, "ms_per_step_max": max(regions) / args.steps * 1e3},
                "roofline": roof}
        return line, (sf, desc, base)

    # ---- path trace: pathtrace.json at 1080p, 1024 spp ------------------------------------------------
    def pathtrace(self):
        args, rt, rtd, torch, np = self.args, self.rt, self.rtd, self.torch, self.np
        spp = args.spp
        sf, desc, base = scene_setup("pathtrace_1080p", spp)
        W, H = base.width, base.height
        side = int(np.sqrt(spp))
        P = rt.copy_params(base)
        P.sample_begin, P.sample_end = rtd.strata_range(self.rank, self.world, side * side)
        ref_counts = self.reference_counts(sf, desc, P)
        r = rt.Renderer(self.local_rank)
        r.upload(desc); r.set_camera(sf.camera); r.set_params(P)
        pipe = rtd.AccumulatorPipeline(self.rank, self.world, H * W, self.dev, depth=2)
        u8 = [torch.zeros(pipe.slice_px * 3, dtype=torch.uint8, device=self.dev) for _ in range(2)]
        stream = torch.cuda.current_stream().cuda_stream
        steps = max(args.pt_steps, 1)

        def enqueue(k, ev=None):
            accum = pipe.target(k)
            if ev: ev[0].record()
            r.accumulate_device(accum.data_ptr(), stream=stream)
            if ev: ev[1].record()
            pipe.submit(k)

        def finish(k):
            # this rank's slice of the summed frame: / pixelSamples, tone map, gamma, clamp, PPM bytes; rank 0 gathers the bytes
            def stages(sums, n_px):
                r.resolve_device(sums.data_ptr(), n_px, stream=stream)
                r.quantize_device(sums.data_ptr(), n_px * 3, u8[k % 2].data_ptr(), stream=stream)
                return u8[k % 2]
            pipe.complete(k, stages)

        def run(n, events=None):
            for k in range(n):
                enqueue(k, events[k] if events else None)
                if k >= 1:
                    finish(k - 1)
            finish(n - 1)

        run(1)  # the first frame sizes the node store (synchronous): not a warm-up step
        self.sync_all()
        run(max(args.pt_warmup, 0)) if args.pt_warmup > 0 else None
        self.sync_all()
        events = [(torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)) for _ in range(steps)]
        t0 = time.perf_counter()
        run(steps, events)
        self.sync_all()
        elapsed = time.perf_counter() - t0
        k_ms = float(np.mean([a.elapsed_time(b) for a, b in events]))
        st = r.stats()
        assert st.rays == ref_counts[0], "the production traversal must trace exactly the reference's rays"
        r.close()

        launches = max(int(st.kernel_launches), 1)
        (elapsed,), cnt, k_ms = self.reduce_counts(ref_counts, [elapsed], k_ms)
        rays, boxes, sph, tri, cyl, unwalked_ops = cnt
        ms_per_step = elapsed / steps * 1e3
        mrays = rays * steps / elapsed / 1e6
        alg_ops = OPS_BOX * boxes + OPS_SPHERE * sph + OPS_TRI * tri + OPS_CYL * cyl  # per frame, all ranks
        achieved = alg_ops / self.world / (k_ms * 1e-3) / 1e12
        walked = (alg_ops - unwalked_ops) / alg_ops
        # Second numerator: what the reference computes per light sample besides the shadow ray's intersection, counted from its
        # expressions like SURVEY 8(d) counts a box test -- AreaLight::sample (AreaLight.cpp:3-20: two offsets 4, point 12,
        # direction 3, length 6, three divisions 3, area 1, cosine 6, pdf 3 = 38), the shadow ray (origin 6, normalize 9 = 15)
        # and, where the sample is neither occluded nor back-facing, the BRDF and the term (RayTracer.cpp:655-679: two cosines 12,
        # diffuse 6, half vector 12, n.h 6, pow 1, specular 12, sum 3, term 12, += 3 = 67).  53 ops for every counted shadow ray
        # is therefore a LOWER bound of the reference's own per-sample fp64 arithmetic (the RNG draws are not priced).
        S = PT_LIGHT_SAMPLES
        shadow_rays = rays * S / (S + 1.0)
        sample_ops = 53.0 * shadow_rays
        roof = {"bound": "valu_fp64", "achieved": achieved, "peak": FP64_VALU_PEAK_TOPS, "unit": "TFLOP/s",
                "frac": achieved / FP64_VALU_PEAK_TOPS, "traffic": None,
                # only the reference-order tests of rays that are walked (not: shadow rays of dark and clear hits)
                "frac_walked": achieved * walked / FP64_VALU_PEAK_TOPS, "walked_ops_share": walked,
                # ... and with the per-sample arithmetic of the reference priced in (lower bound, see bench.py)
                "frac_with_sampling": (alg_ops + sample_ops) / self.world / (k_ms * 1e-3) / 1e12 / FP64_VALU_PEAK_TOPS,
                "sampling_ops_per_shadow_ray": 53.0,
                "kernel": ("pq_kernel (one launch per 1024 strata; every wave on its own: ray pass = 64 stacked rays or new roots -> closest "
                           "hit; hit pass = 64 hits -> class dark / clear / general, material branch, child rays; light pass = 4 hits x "
                           "16 samples; exact per-pixel sums added to the frame)") if os.environ.get("RT_PT_QUEUES", "1") != "0" else
                          "pt_kernel (per batch of 4 strata; per depth: trace, classify, light, shade; combine) + bw_accumulate_path_kernel",
                "kernel_ms": k_ms, "kernel_ms_per_launch": k_ms / launches,
                "launches_per_step": launches, "launches_in_flight": 1, "workgroups_per_launch": "all",
                "ops_per_ray": alg_ops / max(rays, 1.0), "reference_box_tests_per_ray": boxes / max(rays, 1.0),
                "note": "fp64 ops without FMA credit; numerator of `frac` = the REFERENCE's box / primitive tests (SURVEY 8(d)); "
                        "frac_walked drops the tests of rays that are counted but not walked, frac_with_sampling adds the reference's "
                        "per-sample arithmetic"}
        block = {"metric": "spp/s", "value": side * side / (elapsed / steps), "unit": "spp/s at 1920x1080",
                 "mrays_per_s": mrays, "n_gpus": self.world, "steps": steps, "warmup": max(args.pt_warmup, 0),
                 "ms_per_step": ms_per_step, "higher_is_better": True, "scaling": "strong", "dtype": "f64",
                 "data": "reference scene file (scenes/pathtrace.json), resized",
                 "config": {"workload": "pathtrace.json 1920x1080, pixelSamples %d (%dx%d strata), lightSamples %d, nbounces 8"
                                        % (spp, side, side, PT_LIGHT_SAMPLES),
                            "rays_per_step": rays, "parallelism": "strata-range x%d + RCCL reduce-scatter, stages per slice, gather" % self.world},
                 "roofline": roof}
        return block, (sf, desc, base)


def cpu_baseline(sf, desc, base, pathtrace):
    """The pinned oracle (oracle/liboracle.so, -O2 -fopenmp) on this box's host cores: checker code used
    here only as the reported baseline, on a bounded sample of the workload."""
    import raytracer_amd as rt
    from tests import oracle_lib
    threads = min(oracle_lib.load().oracle_max_threads(), os.cpu_count() or 1)
    P = rt.copy_params(base)
    if pathtrace:
        P.width, P.height = 480, 270  # bounded sample: 1/16 of the frame's pixels ...
        sf.resize(480, 270)
        side = int(base.pixel_samples ** 0.5)
        n_strata = min(16, side * side)  # ... and the first 16 of its strata per pass
        P.sample_begin, P.sample_end = 0, n_strata
        sample = "480x270 (1/16 of the pixels), strata 0..%d of %d per pass" % (n_strata - 1, side * side)
    else:
        n_strata = 0
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
        value = n_strata * n / dt / 16.0  # spp/s referred to the full 1080p frame
        unit = "spp/s at 1920x1080"
    else:
        value = rays / dt / 1e6
        unit = "Mrays/s"
    return {"value": value, "unit": unit, "cores": threads, "kind": "port", "mrays_per_s": rays / dt / 1e6,
            "sample": sample + f" ({n} passes, {dt:.1f} s)",
            "note": "our fp64 CPU restatement, byte-identical to the reference on the deterministic scenes; the reference "
                    "itself, as built by its Makefile, measured 0.6-2.1 Mrays/s on 8 cores (BASELINE.md)"}


def launch_command(n_gpus, argv, port):
    """The command that starts one rank per GPU: what the driver itself runs for N > 1 (task brief), with this
    invocation's own arguments handed to every rank."""
    return [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", str(n_gpus),
            "--master-addr", "127.0.0.1", "--master-port", str(port), os.path.abspath(__file__)] + list(argv)


def free_port():
    import socket
    with socket.socket() as s:
        s.bind(("127.0.0.1", 0))
        return s.getsockname()[1]


def visible_gpus():
    import torch
    return torch.cuda.device_count()  # counts devices without initialising the GPU runtime in this process


def launch_ranks(args, argv):
    """`python bench.py --gpus N` outside torch.distributed.run: start the N ranks as a child job BEFORE this process
    makes any GPU call, let rank 0's JSON line through on stdout, return the job's exit code."""
    have = visible_gpus()
    if have < args.gpus:
        print(f"[bench] --gpus {args.gpus} needs {args.gpus} visible GPUs, this box shows {have}: refusing to print a "
              f"{args.gpus}-GPU line from fewer devices", file=sys.stderr)
        return 2
    cmd = launch_command(args.gpus, argv, free_port())
    print("[bench] launching " + " ".join(cmd), file=sys.stderr)
    return subprocess.run(cmd).returncode


def group_bench(args):
    """--mode group: ONE process, devices 0..N-1 behind rt_group_render_u8 (blocking: bands / strata ranges rendered side
    by side, assembled on device 0 by peer copies, the PPM payload copied to the host).  The timed step therefore
    includes the device-to-host copy of the frame's bytes, which the ranks mode (device-resident frames) does not."""
    have = visible_gpus()
    if have < args.gpus:
        print(f"[bench] --mode group --gpus {args.gpus} needs {args.gpus} visible GPUs, this box shows {have}", file=sys.stderr)
        return 2
    import numpy as np
    import raytracer_amd as rt
    line = None
    for workload in ("clutter_500_4k", "pathtrace_1080p"):
        if args.workload not in ("both", workload):
            continue
        pt = workload == "pathtrace_1080p"
        sf, desc, base = scene_setup(workload, args.spp)
        g = rt.Group(list(range(args.gpus)))
        g.load(sf, params=base, desc=desc)
        steps, warmup = (max(args.pt_steps, 1), max(args.pt_warmup, 0)) if pt else (args.steps, args.warmup)
        g.render_u8()  # the members size their node stores: not a warm-up step
        for _ in range(warmup):
            g.render_u8()
        t0 = time.perf_counter()
        for _ in range(steps):
            g.render_u8()  # blocking
        elapsed = time.perf_counter() - t0
        st = g.stats()
        g.close()
        block = {"metric": "spp/s" if pt else "Mrays/s",
                 "value": (int(np.sqrt(args.spp)) ** 2 / (elapsed / steps)) if pt else st.rays * steps / elapsed / 1e6,
                 "unit": "spp/s at 1920x1080" if pt else "Mrays/s", "mrays_per_s": st.rays * steps / elapsed / 1e6,
                 "n_gpus": args.gpus, "steps": steps, "warmup": warmup, "ms_per_step": elapsed / steps * 1e3,
                 "higher_is_better": True, "scaling": "strong", "vs_baseline": None, "dtype": "f64",
                 "data": "reference scene file, resized",
                 "config": {"workload": workload, "rays_per_step": st.rays, "kernel_ms_slowest_member": st.kernel_ms,
                            "parallelism": "rt_group x%d: one process, peer copies to device 0, frame bytes to the host "
                                           "inside the timed step" % args.gpus}}
        if line is None:
            line = block
        else:
            line["pathtrace"] = block
    print(json.dumps(line), flush=True)
    return 0


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1,
                    help="GPUs of this node that share ONE frame; N > 1 without WORLD_SIZE in the environment starts the N "
                         "ranks itself (torch.distributed.run)")
    ap.add_argument("--mode", default="ranks", choices=["ranks", "group"],
                    help="ranks: one process per GPU + RCCL (default); group: ONE process driving devices 0..N-1 through "
                         "rt_group_render_u8 (the C-ABI's own multi-GPU path)")
    ap.add_argument("--steps", type=int, default=20)
    ap.add_argument("--warmup", type=int, default=3)
    ap.add_argument("--workload", default="both", choices=["both", "clutter_500_4k", "pathtrace_1080p"])
    ap.add_argument("--repeat", type=int, default=10, help="Whitted: timed regions of K steps each (the median is the headline)")
    ap.add_argument("--spp", type=int, default=PT_SPP, help="path trace: pixelSamples of a frame (default 1024 = 32x32 strata)")
    ap.add_argument("--pt-steps", type=int, default=2, help="path trace: timed frames")
    ap.add_argument("--pt-warmup", type=int, default=0, help="path trace: untimed frames after the node-store sizing frame")
    ap.add_argument("--in-flight", type=int, default=0,
                    help="Whitted: frames in flight, one render context + stream each (default 4)")
    ap.add_argument("--workgroups", type=int, default=-1,
                    help="workgroups per context (default: 0 = fill the chip when 1 frame is in flight, else 64)")
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--emulate-world", type=int, default=0,
                    help="rehearsal on one GPU: render rank 0's share of an N-rank job (no collective); the printed "
                         "value then refers to that share only")
    ap.add_argument("--counts-child", default=None, help=argparse.SUPPRESS)
    args = ap.parse_args()
    if args.counts_child:
        executed_counts_child(args.counts_child, args.spp)
        return

    if args.gpus < 1:
        ap.error("--gpus must be >= 1")
    if args.mode == "group":
        sys.exit(group_bench(args))
    if "WORLD_SIZE" not in os.environ and args.gpus > 1:
        sys.exit(launch_ranks(args, sys.argv[1:]))
    rank = int(os.environ.get("RANK", "0"))
    world = int(os.environ.get("WORLD_SIZE", "1"))
    if world != args.gpus:
        print(f"[bench] --gpus {args.gpus} but WORLD_SIZE={world}: the launcher and the flag disagree", file=sys.stderr)
        sys.exit(2)
    do_w = args.workload in ("both", "clutter_500_4k")
    do_p = args.workload in ("both", "pathtrace_1080p")
    # child processes first, before this process touches the GPU
    exe_w = executed_counts("clutter_500_4k", args.spp) if (do_w and rank == 0 and world == 1) else None
    exe_p = executed_counts("pathtrace_1080p", args.spp) if (do_p and rank == 0 and world == 1) else None

    b = Bench(args)
    line = None
    if do_w:
        line, setup = b.whitted()
        if rank == 0:
            if exe_w:
                line["roofline"].update(exe_w)
                add_executed_rates(line, line["roofline"], line["config"]["rays_per_step"], line["ms_per_step"])
            q = quoted_profile("r03_clutter_500_pmc.json")
            if q and world == 1:
                line["roofline"]["traffic"] = q.get("hbm_bytes_per_frame")
                line["roofline"]["traffic_fetch_raw"] = q.get("hbm_bytes_per_frame_fetch_raw")
                line["roofline"]["valu_busy"] = q.get("valu_busy")
                line["roofline"]["lane_utilisation"] = q.get("lane_utilisation")
                line["roofline"]["pmc_quoted_from"] = q["quoted_from"] + " (" + q.get("source", "") + ")"
            if not args.no_cpu_baseline and world == 1:
                line["cpu_baseline"] = cpu_baseline(*setup, pathtrace=False)
    if do_p:
        block, setup = b.pathtrace()
        if rank == 0:
            if exe_p:
                block["roofline"].update(exe_p)
                add_executed_rates(block, block["roofline"], block["config"]["rays_per_step"], block["ms_per_step"])
            q = quoted_profile("r03_pathtrace_pmc.json")
            if q and world == 1:
                block["roofline"]["traffic"] = q.get("hbm_bytes_per_launch")
                block["roofline"]["traffic_fetch_raw"] = q.get("hbm_bytes_per_launch_fetch_raw")
                block["roofline"]["traffic_strata_per_launch"] = q.get("strata_per_launch")
                block["roofline"]["valu_busy"] = q.get("valu_busy")
                block["roofline"]["lane_utilisation"] = q.get("lane_utilisation")
                block["roofline"]["pmc_quoted_from"] = q["quoted_from"] + " (" + q.get("source", "") + ")"
            if not args.no_cpu_baseline and world == 1:
                block["cpu_baseline"] = cpu_baseline(*setup, pathtrace=True)
        if line is None:
            block["vs_baseline"] = None
            line = block
        else:
            line["pathtrace"] = block
    if rank == 0:
        print(json.dumps(line), flush=True)
    if world > 1:
        b.dist.barrier()
        b.dist.destroy_process_group()


if __name__ == "__main__":
    main()
