"""Path tracing parity.

(1) Same counter-based RNG on both sides -> the GPU frame must agree with the oracle's almost to
    the bit (cos/sin/pow last-bit differences perturb directions by ~1e-16; a pixel can only differ
    materially if such a perturbation flips a discrete decision, which is allowed for a stated,
    tiny fraction of pixels).
(2) Against the REFERENCE BINARY's committed high-spp renders (tests/golden/pt_ref_*.npy): a
    statistical MSE bound, since the reference seeds from random_device.
(3) Partition invariance: sample ranges and row bands recombine to the one-GPU accumulation.
"""
import glob
import os

import numpy as np
import pytest

import raytracer_amd as rt
from raytracer_amd import capi
from tests import oracle_lib, scene_util

pytestmark = pytest.mark.gpu
GOLDEN = os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden")


def gpu_and_oracle(renderer, sf, P, desc=None):
    desc = desc if desc is not None else sf.desc
    renderer.upload(desc); renderer.set_camera(sf.camera); renderer.set_params(P)
    rgb, _ = renderer.render(want_hit=False)
    st = renderer.stats()
    orgb, _, cnt, _ = oracle_lib.render(desc, sf.camera, P, want_hit=False)
    return rgb, st, orgb, cnt


def assert_pt_parity(rgb, st, orgb, cnt, max_outlier_frac=1e-3):
    diff = np.abs(rgb - orgb).max(axis=2)
    outliers = (diff > 1e-9).mean()
    assert outliers <= max_outlier_frac, f"{outliers:.2e} of the pixels differ by more than 1e-9"
    if outliers == 0:
        assert st.rays == cnt["rays"]
    else:
        assert abs(st.rays - cnt["rays"]) <= 1e-3 * cnt["rays"]


@pytest.mark.parametrize("name,spp,ls", [("pathtrace", 4, 16), ("pathtrace", 16, 2), ("final", 4, 4), ("pt2", 4, 4),
                                         ("lens_sampling", 9, 2), ("indirect_light", 4, 4), ("brdf", 4, 4),
                                         ("pathtrace", 4, 3), ("final", 4, 32), ("final_textured_lens", 4, 4)])
def test_same_rng_same_image(renderer, name, spp, ls):
    """ls = 3 takes the general kernel (bw_kernel<1>), the powers of two pt_kernel; final_textured_lens is BASELINE
    config 5's workload (final.json + textures + thin lens + Uncharted2, scripts/make_config5_scene.py)."""
    sf = scene_util.load(name, 150, 100, tonemap="uncharted2" if name.startswith("final") else None)
    P = sf.params
    P.mode = rt.RT_MODE_PATHTRACE
    P.pixel_samples, P.light_samples = spp, ls
    # atan2 / acos last-bit differences may pick the neighbouring texel at a texel boundary
    assert_pt_parity(*gpu_and_oracle(renderer, sf, P), max_outlier_frac=2e-3 if name == "final_textured_lens" else 1e-3)


def test_general_kernel_equals_the_row_kernel():
    """RT_PT_GENERIC=1 forces bw_kernel<1> where pt_kernel applies: same rays, same image up to the order in which a
    hit's light samples are added."""
    from tests.conftest import _make
    g = _make({"RT_PT_GENERIC": "1"})
    r = _make({"RT_PT_QUEUES": "0"})
    try:
        sf = scene_util.load("final", 120, 80)
        P = sf.params
        P.pixel_samples, P.light_samples = 4, 8
        g.load(sf, P); r.load(sf, P)
        a, _ = g.render(want_hit=False)
        b, _ = r.render(want_hit=False)
        assert np.abs(a - b).max() < 1e-12
        assert g.stats().rays == r.stats().rays
    finally:
        g.close(); r.close()


def test_regrouped_light_sample_equals_the_reference_order():
    """pt_kernel regroups an area-light sample's arithmetic (one reciprocal square root for distance and direction, the
    pdf folded into one weight, no second normalize of the unit direction); RT_PT_EXACT=1 keeps the reference's operation
    order.  Same rays; radiance within a few ulp per sample."""
    from tests.conftest import _make
    e = _make({"RT_PT_EXACT": "1"})  # (both through pq_kernel, which shares the light sample's code with pt_kernel)
    r = _make({})
    try:
        for name, spp, ls in (("pathtrace", 16, 16), ("final", 4, 8)):
            sf = scene_util.load(name, 150, 100)
            P = sf.params
            P.pixel_samples, P.light_samples = spp, ls
            e.load(sf, P); r.load(sf, P)
            a, _ = e.render(want_hit=False)
            b, _ = r.render(want_hit=False)
            assert e.stats().rays == r.stats().rays
            assert np.abs(a - b).max() < 1e-12
            assert not np.array_equal(a, b)   # (the knob does something)
    finally:
        e.close(); r.close()


@pytest.mark.parametrize("queues", ["1", "0"])
def test_hit_classes_change_no_bit(queues):
    """(queues = 1: pq_kernel, the per-wave queues; 0: pt_kernel, the workgroup-wide levels -- both classify.)
    pt_kernel sorts a level's hits into dark (the whole light behind the surface or the hit behind the light: direct
    light exactly 0), clear (the shaft from the hit to the light misses every other primitive's enlarged box: no any-hit
    walks) and general before its light phase.  Both shortcuts are exact: RT_PT_CLASSES=0 (every hit general) and
    RT_PT_CLASSES=1 (dark only) must give the same bits and the same ray count as the default, on the shipped scenes
    (open, closed box, textured, thin lens) and on random soups of all three primitive types -- glass and mirrors among
    them -- under randomly placed, tilted, large and small area lights."""
    from tests.conftest import _make
    rs = [_make({"RT_PT_CLASSES": "0", "RT_PT_QUEUES": queues}), _make({"RT_PT_CLASSES": "1", "RT_PT_QUEUES": queues}),
          _make({"RT_PT_QUEUES": queues})]
    try:
        cases = []
        for name, spp, ls in (("pathtrace", 4, 16), ("final", 4, 8), ("final_textured_lens", 4, 4), ("lens_sampling", 4, 2),
                              ("indirect_light", 4, 4), ("brdf", 4, 4)):
            sf = scene_util.load(name, 150, 100)
            P = sf.params
            P.mode = rt.RT_MODE_PATHTRACE
            P.pixel_samples, P.light_samples = spp, ls
            cases.append((name, sf.desc, sf.camera, P, sf))
        rng = np.random.default_rng(77)
        mats = [scene_util.default_material(), scene_util.default_material(reflective=1, reflectivity=0.8),
                scene_util.default_material(refractive=1, ior=1.5), scene_util.default_material(diffuse=(0.2, 0.9, 0.3), exp=5)]
        for k in range(10):
            n_sph, n_tri, n_cyl = [(6, 6, 4), (40, 30, 20), (1, 0, 0), (0, 2, 1), (150, 100, 50)][k % 5]
            types, geom = scene_util.random_prims(rng, n_sph, n_tri, n_cyl, extent=3.0)
            nrm = rng.standard_normal(3)
            nrm /= np.linalg.norm(nrm)
            u = np.cross(nrm, rng.standard_normal(3))
            u /= np.linalg.norm(u)
            v = np.cross(nrm, u)
            size = [0.02, 0.5, 2.0, 6.0][k % 4]          # a thin shaft ... a light larger than the scene
            pos = rng.uniform(-3, 3, 3) if k % 3 else rng.uniform(-12, 12, 3)  # inside the soup / outside it
            if k % 2:  # a `normal` that is not perpendicular to u and v, and u, v that are not perpendicular to each other
                nrm = nrm + 0.7 * u
                nrm /= np.linalg.norm(nrm)
                v = v + 0.5 * u
                v /= np.linalg.norm(v)
            light = scene_util.area_light(pos, nrm, u, v, size, size * rng.uniform(0.3, 1.0))
            syn = scene_util.SyntheticScene(types, geom, materials=mats, lights=[light], use_bvh=k % 2,
                                            prim_material=rng.integers(0, len(mats), len(types)))
            cam = scene_util.look_at(rng.uniform(-6, 6, 3), (0, 0, 0), aspect=1.5)
            P = scene_util.make_params(120, 80, mode=rt.RT_MODE_PATHTRACE, max_depth=6, pixel_samples=4, light_samples=[4, 16, 1, 8][k % 4], seed=k)
            cases.append((f"soup{k}", syn.desc, cam, P, syn))
        for name, desc, cam, P, keep in cases:
            out = []
            for r in rs:
                r.upload(desc); r.set_camera(cam); r.set_params(P)
                rgb, _ = r.render(want_hit=False)
                out.append((rgb, r.stats().rays))
            for rgb, rays in out[:2]:
                assert rays == out[2][1], name
                assert np.array_equal(rgb, out[2][0]), name
    finally:
        for r in rs:
            r.close()


def test_two_lights_take_the_general_kernel(renderer):
    """A point light beside the area light (S = 1 + 4 shadow rays per hit): not pt_kernel's case."""
    sf = scene_util.load("pathtrace", 90, 60)
    P = sf.params
    P.pixel_samples, P.light_samples = 4, 4
    d = sf.desc
    lights = (capi.Light * 2)(d.lights[0], scene_util.point_light((1.0, 2.0, -1.0), (0.6, 0.5, 0.4)))
    import ctypes as C
    desc = capi.SceneDesc()
    C.memmove(C.byref(desc), C.byref(d), C.sizeof(capi.SceneDesc))
    desc.lights = C.cast(lights, C.POINTER(capi.Light))
    desc.n_lights = 2
    assert_pt_parity(*gpu_and_oracle(renderer, sf, P, desc=desc))


def test_headline_configuration_at_full_size(renderer):
    """BASELINE config 4 as bench.py runs it: pathtrace.json at 1920x1080, pixelSamples 1024 (32x32 strata), lightSamples
    16.  The oracle cannot render 2 x 10^9 pixel-samples in a test, so: one stratum of the full-size frame against the
    oracle (same rays exactly, same radiance sums), and size-independent properties of the whole frame -- finite,
    inside [0, 1], and as many rays as its 1024 strata predict within the spread of one stratum."""
    import torch
    sf = scene_util.load("pathtrace", 1920, 1080)
    P = sf.params
    P.pixel_samples, P.light_samples = 1024, 16
    stream = torch.cuda.current_stream().cuda_stream
    acc = torch.empty(1080 * 1920 * 3, dtype=torch.float64, device="cuda")
    Q = rt.copy_params(P)
    Q.sample_begin, Q.sample_end = 517, 518          # one stratum out of the middle
    renderer.load(sf, Q)
    renderer.accumulate_device(acc.data_ptr(), stream=stream)
    torch.cuda.synchronize()
    st = renderer.stats()
    _, _, cnt, oacc = oracle_lib.render(sf.desc, sf.camera, Q, want_hit=False, want_accum=True)
    assert st.rays == cnt["rays"]
    got = acc.cpu().numpy().reshape(1080, 1920, 3)
    assert (np.abs(got - oacc).max(axis=2) > 1e-9).mean() <= 1e-4
    one_stratum = st.rays
    renderer.set_params(P)
    rgb, _ = renderer.render(want_hit=False)
    assert np.isfinite(rgb).all() and rgb.min() >= 0.0 and rgb.max() <= 1.0
    assert abs(renderer.stats().rays / 1024 - one_stratum) < 0.01 * one_stratum
    assert 0.3 < rgb.mean() < 0.9   # sky-lit scene, gamma applied


def test_config5_workload_at_full_size(renderer):
    """BASELINE config 5's workload at its own size: final_textured_lens.json (final.json + textures + thin lens) at
    3840x2160 with pixelSamples 4096 (64x64 strata), lightSamples 16.  One stratum of the 4K frame against the oracle
    (same rays exactly; radiance sums to 1e-9 but for the few pixels where an atan2 / acos last bit picks the neighbouring
    texel), then eight strata: finite, non-negative, and as many rays as one stratum predicts."""
    import torch
    W, H = 3840, 2160
    sf = scene_util.load("final_textured_lens", W, H, tonemap="uncharted2")
    P = sf.params
    P.pixel_samples, P.light_samples = 4096, 16
    stream = torch.cuda.current_stream().cuda_stream
    acc = torch.empty(H * W * 3, dtype=torch.float64, device="cuda")
    Q = rt.copy_params(P)
    Q.sample_begin, Q.sample_end = 2077, 2078
    renderer.load(sf, Q)
    renderer.accumulate_device(acc.data_ptr(), stream=stream)
    torch.cuda.synchronize()
    st = renderer.stats()
    _, _, cnt, oacc = oracle_lib.render(sf.desc, sf.camera, Q, want_hit=False, want_accum=True)
    got = acc.cpu().numpy().reshape(H, W, 3)
    outliers = (np.abs(got - oacc).max(axis=2) > 1e-9).mean()
    assert outliers <= 2e-3
    assert abs(st.rays - cnt["rays"]) <= (0 if outliers == 0 else 1e-3 * cnt["rays"])
    one_stratum = st.rays
    Q.sample_begin, Q.sample_end = 0, 8
    renderer.set_params(Q)
    renderer.accumulate_device(acc.data_ptr(), stream=stream)
    torch.cuda.synchronize()
    got = acc.cpu().numpy()
    assert np.isfinite(got).all() and got.min() >= 0.0
    assert abs(renderer.stats().rays / 8 - one_stratum) < 0.01 * one_stratum


@pytest.mark.parametrize("which", ["levels", "mega", "fused_whole"])
def test_other_pipelines_path_trace_the_same_image(request, which):
    r = request.getfixturevalue("renderer_" + which)
    sf = scene_util.load("final", 120, 80)
    P = sf.params
    P.pixel_samples, P.light_samples = 4, 3
    assert_pt_parity(*gpu_and_oracle(r, sf, P))


def test_scene_too_large_for_lds_path_traced(renderer):
    """1900 primitives: the kernels read the scene image from HBM instead of a staged LDS copy."""
    from tests.test_gpu_parity import large_soup
    types, geom, mats, pm, lights, cam = large_soup()
    s = scene_util.SyntheticScene(types, geom, materials=mats, lights=lights, prim_material=pm, use_bvh=1)
    P = scene_util.make_params(96, 64, mode=rt.RT_MODE_PATHTRACE, max_depth=5, pixel_samples=4, light_samples=2)
    renderer.upload(s.desc); renderer.set_camera(cam); renderer.set_params(P)
    rgb, _ = renderer.render(want_hit=False)
    st = renderer.stats()
    assert st.scene_in_lds == 0
    orgb, _, cnt, _ = oracle_lib.render(s.desc, cam, P, want_hit=False)
    assert_pt_parity(rgb, st, orgb, cnt)


def test_many_strata_in_several_passes(renderer, monkeypatch):
    """A node budget small enough to force several passes over the strata: sums must not change."""
    sf = scene_util.load("pathtrace", 64, 48)
    P = sf.params
    P.pixel_samples, P.light_samples = 36, 2
    renderer.load(sf, P)
    a, _ = renderer.render(want_hit=False)
    monkeypatch.setenv("RT_WF_NODE_BUDGET", "70000")
    P.seed = 1
    renderer.set_params(P)   # new generation -> re-plan with the small budget
    P.seed = 0
    renderer.set_params(P)
    b, _ = renderer.render(want_hit=False)
    assert np.array_equal(a, b)
    orgb, _, _, _ = oracle_lib.render(sf.desc, sf.camera, P, want_hit=False)
    assert (np.abs(b - orgb).max(axis=2) > 1e-9).mean() <= 1e-3


def test_non_square_sample_count_darkens_like_the_reference(renderer):
    # pixelSamples = 10 -> 3x3 strata but divided by 10 (RayTracer.cpp:255, 301)
    sf = scene_util.load("pathtrace", 90, 60)
    P = sf.params
    P.pixel_samples, P.light_samples = 10, 2
    assert_pt_parity(*gpu_and_oracle(renderer, sf, P))


def test_textured_lens_uncharted2_variant(renderer):
    """BASELINE config 5's feature set on final.json: texture + aperture + Uncharted2."""
    sf = scene_util.load("textured", 150, 100, tonemap="uncharted2")
    desc = sf.desc
    P = sf.params
    P.mode = rt.RT_MODE_PATHTRACE
    P.pixel_samples, P.light_samples = 4, 2
    lens = scene_util.load("lens_sampling", 150, 100)
    renderer.upload(desc); renderer.set_camera(lens.camera); renderer.set_params(P)

    class Both:
        camera = lens.camera
    rgb, _ = renderer.render(want_hit=False)
    orgb, _, _, _ = oracle_lib.render(desc, lens.camera, P, want_hit=False)
    diff = np.abs(rgb - orgb).max(axis=2)
    # atan2/acos last-bit differences may pick the neighbouring texel at a texel boundary
    assert (diff > 1e-9).mean() <= 2e-3


def test_different_seeds_differ_and_converge(renderer):
    sf = scene_util.load("pathtrace", 90, 60)
    P = sf.params
    P.pixel_samples, P.light_samples = 16, 4
    renderer.load(sf, P)
    a, _ = renderer.render(want_hit=False)
    P.seed = 12345
    renderer.set_params(P)
    b, _ = renderer.render(want_hit=False)
    assert not np.array_equal(a, b)
    assert np.mean((a - b) ** 2) < 2e-3


@pytest.mark.parametrize("fixture", sorted(glob.glob(os.path.join(GOLDEN, "pt_ref_*.npy"))) or ["<none>"])
def test_mse_against_reference_binary_render(renderer, fixture):
    """The committed images were rendered by the unmodified reference (oracle/_ref) with many
    samples (1024 spp x 16 light samples for pathtrace.json and for config 5's final_textured_lens.json, 256 x 4 for
    final.json).  The GPU renders the same configuration with its own RNG; in the 8-bit output domain the mean
    squared error must be what two independent Monte-Carlo estimates of one image give, and the per-channel
    mean difference must vanish.  Measured (3 seeds each, scripts/gpu_pt_stats.py): MSE 2.0e-6 .. 4.3e-6, bias
    <= 2.5e-5.  Bounds: MSE < 2e-5, bias < 1e-4 -- a 1 % error in the direct light alone would shift the mean by
    ~1e-3 (values ~0.7 after gamma)."""
    if fixture == "<none>":
        pytest.skip("no reference path-trace fixtures committed")
    base = os.path.basename(fixture)[len("pt_ref_"):-len(".npy")]
    name, dims, spp, ls = base.rsplit("_", 3)
    w, h = (int(x) for x in dims.split("x"))
    spp, ls = int(spp[:-3]), int(ls[:-2])
    ref = np.load(fixture).astype(np.float64) / 255.0  # [h, w, 3], top row first
    sf = scene_util.load(name, w, h, tonemap="uncharted2" if name.startswith("final") else None)
    P = sf.params
    P.pixel_samples, P.light_samples = spp, ls
    renderer.load(sf, P)
    rgb, _ = renderer.render(want_hit=False)
    img = np.floor(255.999 * rgb[::-1]) / 255.0
    mse = np.mean((img - ref) ** 2)
    bias = np.abs((img - ref).mean(axis=(0, 1)))
    assert mse < 2e-5, mse
    assert bias.max() < 1e-4, bias


def test_sample_range_and_band_partitions_recombine(renderer):
    """pathtrace split across GPUs: per-rank accumulations of disjoint strata ranges sum to the
    one-GPU accumulation (fp add order differs -> 1e-12 relative), and resolve() of the sum equals
    the one-call render."""
    import ctypes as C
    import torch
    sf = scene_util.load("pathtrace", 96, 64)
    P = sf.params
    P.pixel_samples, P.light_samples = 16, 2
    renderer.load(sf, P)
    full, _ = renderer.render(want_hit=False)
    n_px = 96 * 64
    total = torch.zeros(n_px * 3, dtype=torch.float64, device="cuda")
    part = torch.empty_like(total)
    stream = torch.cuda.current_stream().cuda_stream
    for lo, hi in [(0, 5), (5, 6), (6, 16)]:
        Q = sf.params
        Q.pixel_samples, Q.light_samples = 16, 2
        Q.sample_begin, Q.sample_end = lo, hi
        renderer.set_params(Q)
        renderer.accumulate_device(part.data_ptr(), stream=stream)
        total += part
    renderer.set_params(P)
    renderer.accumulate_device(part.data_ptr(), stream=stream)
    torch.cuda.synchronize()
    assert torch.allclose(total, part, rtol=1e-12, atol=1e-14)
    renderer.resolve_device(total.data_ptr(), n_px, stream=stream)
    torch.cuda.synchronize()
    got = total.cpu().numpy().reshape(64, 96, 3)
    assert np.abs(got - full).max() < 1e-12
    # bands
    got = np.zeros_like(full)
    for rank in range(3):
        Q = sf.params
        Q.pixel_samples, Q.light_samples = 16, 2
        Q.band_rank, Q.band_world = rank, 3
        renderer.set_params(Q)
        rows = [j for j in range(64) if (j // 8) % 3 == rank]
        got[rows], _ = renderer.render(want_hit=False)
    assert np.array_equal(got, full)


@pytest.mark.parametrize("queues", ["1", "0"])
@pytest.mark.parametrize("gap", [5e-5, 2e-5, 9e-5, -5e-5])
def test_occluder_within_the_bias_of_the_light(queues, gap):
    """A shadow segment starts bias = 1e-4 off the surface along the normal and keeps the length |X - P|: it ends bias
    BEYOND the light point X (RayTracer.cpp:649-651).  A ceiling `gap` < bias behind a downward-facing light therefore
    shadows every sample of an upward-facing floor in the reference (gap < 0: a sheet just in front of the light does so
    anyway); a wall beside the light is grazed by the shifted segments only.  The "clear" class must see both: frames with
    and without the hit classes agree bit for bit, and with the oracle."""
    from tests import oracle_lib
    from tests.conftest import _make
    T = rt.capi.RT_TRIANGLE
    y = 1.5 + gap
    tris = [((-3, -0.5, -3), (3, -0.5, -3), (3, -0.5, 3)), ((-3, -0.5, -3), (3, -0.5, 3), (-3, -0.5, 3)),   # floor
            ((-2, y, -2), (2, y, -2), (0, y, 3)),                                                                # ceiling behind the light
            ((0.5 + gap, 0.0, -2), (0.5 + gap, 3.0, -2), (0.5 + gap, 1.5, 3))]                                   # wall beside it
    geom = np.zeros((len(tris), rt.capi.RT_GEOM_STRIDE))
    for k, (a, b, c) in enumerate(tris):
        geom[k, 0:3], geom[k, 3:6], geom[k, 6:9] = a, b, c
    light = scene_util.area_light((0.0, 1.5, 0.3), (0, -1, 0), (1, 0, 0), (0, 0, 1), 1.0, 1.0)
    syn = scene_util.SyntheticScene(np.full(len(tris), T, dtype=np.int32), geom, materials=[scene_util.default_material()], lights=[light], use_bvh=1)
    cam = scene_util.look_at((-2.0, 1.0, 0.0), (0.0, -0.1, 1.0), aspect=1.5)
    P = scene_util.make_params(120, 80, mode=rt.RT_MODE_PATHTRACE, max_depth=3, pixel_samples=4, light_samples=16)
    out = []
    for classes in ("2", "0"):
        r = _make({"RT_PT_CLASSES": classes, "RT_PT_QUEUES": queues})
        try:
            r.upload(syn.desc); r.set_camera(cam); r.set_params(P)
            rgb, _ = r.render(want_hit=False)
            out.append((rgb, r.stats().rays))
        finally:
            r.close()
    assert out[0][1] == out[1][1]
    assert np.array_equal(out[0][0], out[1][0])
    orgb, _, cnt, _ = oracle_lib.render(syn.desc, cam, P, want_hit=False)
    assert cnt["rays"] == out[0][1]
    assert (np.abs(out[0][0] - orgb).max(axis=2) > 1e-9).mean() <= 1e-3
