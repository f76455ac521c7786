"""pq_kernel (pathqueue_kernels.inc): path tracing out of per-wave queues, the production path for scenes that fit in LDS.

Against the oracle it is covered by every test of test_gpu_pathtrace.py (it is the default).  Here: what is specific to
it -- the level-synchronous pt_kernel must agree with it (same rays; radiance to the order in which a pixel's terms are
added), its exact fixed-point sums make a frame independent of how the roots are dealt to jobs, and the accumulator
arithmetic itself."""
import os
from fractions import Fraction

import numpy as np
import pytest

import raytracer_amd as rt
from tests import scene_util
from tests.conftest import _make

pytestmark = pytest.mark.gpu

CASES = [("pathtrace", 150, 100, 16, 16), ("final", 120, 80, 4, 8), ("final_textured_lens", 120, 80, 4, 4),
         ("lens_sampling", 90, 60, 9, 2), ("pathtrace", 97, 61, 5, 1), ("pathtrace", 64, 48, 100, 4)]


def load(name, w, h, spp, ls):
    sf = scene_util.load(name, w, h, tonemap="uncharted2" if name.startswith("final") else None)
    P = sf.params
    P.mode = rt.RT_MODE_PATHTRACE
    P.pixel_samples, P.light_samples = spp, ls
    return sf, P


def test_level_kernel_and_queue_kernel_agree():
    """RT_PT_QUEUES=0 selects pt_kernel (workgroup-wide levels, values combined bottom-up as the reference does) where
    pq_kernel (per-wave queues, exact sum of T x term per pixel) applies: same ray count, radiance sums 1e-12 apart, odd
    image sizes, non-square sample counts and 1 ... 16 light samples included."""
    lv, qu = _make({"RT_PT_QUEUES": "0"}), _make({})
    try:
        for name, w, h, spp, ls in CASES:
            sf, P = load(name, w, h, spp, ls)
            lv.load(sf, P); qu.load(sf, P)
            a, _ = lv.render(want_hit=False)
            b, _ = qu.render(want_hit=False)
            assert lv.stats().rays == qu.stats().rays, name
            assert np.abs(a - b).max() < 1e-12, name
            assert not np.array_equal(a, b), name  # (two different kernels did run)
    finally:
        lv.close(); qu.close()


def test_frame_does_not_depend_on_how_roots_are_dealt(monkeypatch):
    """A pixel's terms are summed exactly (128-bit fixed point) and rounded once: job size, the golden-ratio dealing order,
    the cost-ordered job table, the frame's repetition and the kernel's instantiation (12 or 16 waves per workgroup) must
    not change a bit."""
    r = _make({})
    try:
        for name, w, h, spp, ls in CASES[:3]:
            sf, P = load(name, w, h, spp, ls)
            r.load(sf, P)
            ref, _ = r.render(want_hit=False)
            rays = r.stats().rays
            again, _ = r.render(want_hit=False)  # (the second frame of a setup follows the first one's job costs)
            assert np.array_equal(ref, again)
            for env in ({"RT_PQ_JOB": "64"}, {"RT_PQ_JOB": "2048"}, {"RT_PQ_NO_INTERLEAVE": "1"}, {"RT_PQ_NO_ORDER": "1"},
                        {"RT_PQ_MIN_PIECE": "8"}, {"RT_PQ_WAVES": "12"}, {"RT_PQ_WAVES": "16"}):
                for k, v in env.items():
                    monkeypatch.setenv(k, v)
                for _ in range(2):
                    got, _ = r.render(want_hit=False)
                    assert np.array_equal(ref, got), (name, env)
                    assert r.stats().rays == rays
                for k in env:
                    monkeypatch.delenv(k)
    finally:
        r.close()


def test_strata_per_launch_change_rounding_only(monkeypatch):
    """RT_PQ_BATCH: strata per launch.  Within a launch a pixel's strata are summed exactly, launches are added in fp64."""
    r = _make({})
    try:
        sf, P = load("pathtrace", 96, 64, 36, 4)
        r.load(sf, P)
        ref, _ = r.render(want_hit=False)
        for batch in ("1", "5", "36"):
            monkeypatch.setenv("RT_PQ_BATCH", batch)
            got, _ = r.render(want_hit=False)
            assert np.abs(got - ref).max() < 1e-13
    finally:
        r.close()


def test_many_bounces_and_glass_everywhere(renderer):
    """Ray trees as wide as they get: nothing but glass and mirrors around the camera, maxDepth 12 (the ray stack's bound is
    64 + 128 x (maxDepth + 2) entries) -- against the oracle."""
    from tests import oracle_lib
    rng = np.random.default_rng(5)
    types, geom = scene_util.random_prims(rng, 12, 6, 3, extent=2.0)
    shell = np.zeros((1, geom.shape[1]))
    shell[0, 3] = 3.5  # a glass ball around the camera and everything else: no ray leaves without refracting
    types, geom = np.concatenate([np.array([rt.capi.RT_SPHERE], dtype=np.int32), types]), np.concatenate([shell, geom])
    mats = [scene_util.default_material(refractive=1, ior=1.5), scene_util.default_material(reflective=1, reflectivity=0.9),
            scene_util.default_material(refractive=1, ior=1.33, diffuse=(0.9, 0.9, 0.9))]
    light = scene_util.area_light((0.0, 4.0, 0.0), (0, -1, 0), (1, 0, 0), (0, 0, 1), 2.0, 2.0)
    pm = rng.integers(0, 3, len(types))
    pm[0] = 0
    syn = scene_util.SyntheticScene(types, geom, materials=mats, lights=[light], use_bvh=1, prim_material=pm)
    cam = scene_util.look_at((0.3, 0.2, 0.1), (1, 0.5, 2), aspect=1.5)
    P = scene_util.make_params(96, 64, mode=rt.RT_MODE_PATHTRACE, max_depth=12, pixel_samples=4, light_samples=4, seed=3)
    renderer.upload(syn.desc); renderer.set_camera(cam); renderer.set_params(P)
    rgb, _ = renderer.render(want_hit=False)
    st = renderer.stats()
    orgb, _, cnt, _ = oracle_lib.render(syn.desc, cam, P, want_hit=False)
    assert (np.abs(rgb - orgb).max(axis=2) > 1e-9).mean() <= 2e-3
    assert abs(st.rays - cnt["rays"]) <= 1e-3 * cnt["rays"]
    assert st.rays > 20 * 96 * 64 * 4  # (the trees are wide)


def test_exact_accumulator_arithmetic(renderer):
    """rt_debug_math op 14: a + b + 3a - b through pq_fixed / pq_acc_value.  Terms are truncated towards zero to multiples
    of 2^-80, added as integers and rounded to nearest-even once."""
    rng = np.random.default_rng(11)
    n = 20000
    a = np.ldexp(rng.uniform(0.5, 1.0, n), rng.integers(-86, 44, n)) * rng.choice([-1.0, 1.0], n)
    b = np.ldexp(rng.uniform(0.5, 1.0, n), rng.integers(-86, 44, n)) * rng.choice([-1.0, 1.0], n)
    a[:8] = [0.0, 1.0, -1.0, 2.0 ** -80, 2.0 ** -81, 1.0 + 2.0 ** -52, 3.0, 2.0 ** 44]  # (the accumulator holds |sum| < 2^47)
    b[:8] = [0.0, 2.0 ** -53, 2.0 ** -53, 2.0 ** -80, -(2.0 ** -80), 2.0 ** -60, -3.0, 2.0 ** -80]
    got = renderer.math(14, a, b)
    one = Fraction(2) ** 80

    def trunc(x):  # towards zero to a multiple of 2^-80
        f = Fraction(float(x)) * one
        return Fraction(int(f)) / one   # int() truncates towards zero

    for k in range(n):
        exact = trunc(a[k]) + trunc(b[k]) + trunc(3.0 * a[k]) + trunc(-b[k])
        assert got[k] == float(exact), (k, a[k], b[k], got[k], float(exact))  # float(Fraction) rounds to nearest-even


def test_mid_sized_scene_takes_the_twelve_wave_kernel(renderer, monkeypatch):
    """90 primitives: the image leaves 16 waves fewer than 152 accumulator slots each, so pq_kernel runs with 12 waves per
    workgroup (and fewer slots than the tiny scenes get) -- oracle parity, and the level kernel agrees."""
    from tests import oracle_lib
    rng = np.random.default_rng(21)
    types, geom = scene_util.random_prims(rng, 50, 30, 10, extent=3.0)
    mats = [scene_util.default_material(), scene_util.default_material(reflective=1, reflectivity=0.8),
            scene_util.default_material(refractive=1, ior=1.5)]
    light = scene_util.area_light((0.0, 5.0, 0.0), (0, -1, 0), (1, 0, 0), (0, 0, 1), 1.5, 1.5)
    syn = scene_util.SyntheticScene(types, geom, materials=mats, lights=[light], use_bvh=1, prim_material=rng.integers(0, 3, len(types)))
    cam = scene_util.look_at((5, 3, -6), (0, 0, 0), aspect=1.5)
    P = scene_util.make_params(96, 64, mode=rt.RT_MODE_PATHTRACE, max_depth=5, pixel_samples=4, light_samples=16)
    renderer.upload(syn.desc); renderer.set_camera(cam); renderer.set_params(P)
    rgb, _ = renderer.render(want_hit=False)
    st = renderer.stats()
    assert st.kernel_launches == 1  # (pq_kernel: one launch for the 4 strata; pt_kernel would also report 1 here, hence the next check)
    orgb, _, cnt, _ = oracle_lib.render(syn.desc, cam, P, want_hit=False)
    assert (np.abs(rgb - orgb).max(axis=2) > 1e-9).mean() <= 1e-3
    assert abs(st.rays - cnt["rays"]) <= 1e-3 * cnt["rays"]
    lv = _make({"RT_PT_QUEUES": "0"})
    try:
        lv.upload(syn.desc); lv.set_camera(cam); lv.set_params(P)
        a, _ = lv.render(want_hit=False)
        assert lv.stats().rays == st.rays
        assert np.abs(a - rgb).max() < 1e-12 and not np.array_equal(a, rgb)
    finally:
        lv.close()


def test_scene_too_large_for_the_queues_takes_the_level_kernel(renderer):
    """pq_kernel needs the scene image and 136 root slots per wave in LDS; a 300-primitive soup leaves no room and is path
    traced by pt_kernel (still with the image in LDS) -- same oracle parity."""
    from tests import oracle_lib
    rng = np.random.default_rng(9)
    types, geom = scene_util.random_prims(rng, 150, 100, 50, extent=3.0)
    light = scene_util.area_light((0.0, 5.0, 0.0), (0, -1, 0), (1, 0, 0), (0, 0, 1), 1.0, 1.0)
    syn = scene_util.SyntheticScene(types, geom, materials=[scene_util.default_material()], lights=[light], use_bvh=1)
    cam = scene_util.look_at((5, 3, -6), (0, 0, 0), aspect=1.5)
    P = scene_util.make_params(96, 64, mode=rt.RT_MODE_PATHTRACE, max_depth=5, pixel_samples=4, light_samples=4)
    renderer.upload(syn.desc); renderer.set_camera(cam); renderer.set_params(P)
    rgb, _ = renderer.render(want_hit=False)
    st = renderer.stats()
    assert st.scene_in_lds == 1
    orgb, _, cnt, _ = oracle_lib.render(syn.desc, cam, P, want_hit=False)
    assert (np.abs(rgb - orgb).max(axis=2) > 1e-9).mean() <= 1e-3


def test_a_ray_stack_that_runs_out_hands_the_setup_to_the_level_kernel(monkeypatch):
    """pq_kernel's stack bound excludes an overflow; RT_PQ_DEBUG_STACK makes the stacks too small on purpose.  The wave that
    would write past its stack sets the flag instead, the blocking entry point discards the frame and renders it again --
    with pt_kernel, which this context then keeps for the setup: the frame a RT_PT_QUEUES=0 context renders, bit for bit."""
    sf, P = load("pathtrace", 120, 80, 16, 4)
    lv = _make({"RT_PT_QUEUES": "0"})
    monkeypatch.setenv("RT_PQ_DEBUG_STACK", "80")
    r = _make({})
    try:
        lv.load(sf, P)
        want, _ = lv.render(want_hit=False)
        r.load(sf, P)
        got, _ = r.render(want_hit=False)
        assert np.array_equal(got, want)
        assert r.stats().rays == lv.stats().rays
        again, _ = r.render(want_hit=False)  # no second overflow: the queues are off for this context
        assert np.array_equal(again, want)
        # the asynchronous entry point reports instead of repeating: first call renders an invalid frame, the next one says so
        r2 = _make({})
        try:
            import torch
            r2.load(sf, P)
            acc = torch.zeros(80 * 120 * 3, dtype=torch.float64, device="cuda")
            stream = torch.cuda.current_stream().cuda_stream
            r2.accumulate_device(acc.data_ptr(), stream=stream)
            torch.cuda.synchronize()
            with pytest.raises(rt.RtError, match="overflowed"):
                r2.accumulate_device(acc.data_ptr(), stream=stream)
            r2.accumulate_device(acc.data_ptr(), stream=stream)  # pt_kernel now
            torch.cuda.synchronize()
            r2.stats()
        finally:
            r2.close()
    finally:
        lv.close(); r.close()
