"""GPU parity proper: the HIP render loop, called through the C-ABI, against the oracle on the same
inputs, against the committed golden hit-ID maps, and against the reference's own PPM hashes.

Bars (BASELINE.json north_star): primary-hit primitive IDs bit-exact; pixels within 1e-4 relative.
What is actually asserted is much tighter: every arithmetic step is IEEE-exact on both sides
(division and sqrt are correctly rounded on gfx950), so colours may differ only where the device
pow() differs from glibc's in the last bit -- a few 1e-16.
"""
import hashlib
import os

import numpy as np
import pytest

import raytracer_amd as rt

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
from tests import oracle_lib, scene_util

pytestmark = pytest.mark.gpu

REL_TOL = 1e-4     # the contract
TIGHT_ABS = 1e-12  # what the design delivers for Whitted scenes


def run_both(renderer, sf, desc=None, params=None):
    desc = desc if desc is not None else sf.desc
    P = params if params is not None else sf.params
    renderer.upload(desc)
    renderer.set_camera(sf.camera)
    renderer.set_params(P)
    rgb, hit = renderer.render()
    st = renderer.stats()
    orgb, ohit, cnt, _ = oracle_lib.render(desc, sf.camera, P)
    return rgb, hit, st, orgb, ohit, cnt


def assert_frame_parity(rgb, hit, st, orgb, ohit, cnt, check_hit=True, check_counts=False):
    if check_hit:
        assert np.array_equal(hit, ohit), f"{(hit != ohit).sum()} primary hit IDs differ"
    assert st.rays == cnt["rays"]
    if check_counts:  # only the visit-for-visit traversal modes make the reference's number of tests
        assert st.box_tests == cnt["boxes"] and st.prim_tests == cnt["prims"]
    diff = np.abs(rgb - orgb)
    assert diff.max() <= TIGHT_ABS, f"max abs colour difference {diff.max()}"
    assert (diff <= REL_TOL * np.maximum(np.abs(orgb), 1e-12)).all()


SMALL = ["binary_primitives", "blinn_phong", "mirror_glass_cylinder", "textured", "simple_phong", "tone_mapping",
         "glass", "mirror_image", "perspective", "scene", "umbrella", "clutter_100"]


@pytest.mark.parametrize("name", SMALL)
def test_whitted_scenes_300x200(renderer, name, golden_hits):
    sf = scene_util.load(name, 300, 200)
    P = sf.params
    if P.mode == rt.RT_MODE_PATHTRACE:
        P.mode = rt.RT_MODE_PHONG  # area lights then act as a point at the origin (Light.cpp:9-11)
    out = run_both(renderer, sf, params=P)
    assert_frame_parity(*out)
    key = f"{name}:bvh1"
    if key in golden_hits:
        assert np.array_equal(out[1], golden_hits[key].astype(np.int32))


@pytest.mark.parametrize("name", ["clutter_500", "mirror_glass_cylinder", "final", "textured"])
def test_level_kernel_pipeline_agrees(renderer_levels, name):
    sf = scene_util.load(name, 240, 160)
    P = sf.params
    if P.mode == rt.RT_MODE_PATHTRACE:
        P.mode = rt.RT_MODE_PHONG
    assert_frame_parity(*run_both(renderer_levels, sf, params=P))


@pytest.mark.parametrize("name", ["clutter_500", "mirror_glass_cylinder", "binary_primitives", "final", "textured"])
@pytest.mark.parametrize("which", ["walk", "mega", "literal"])
def test_other_gpu_implementations_agree_and_count_like_the_reference(request, which, name):
    """The exact-walk wavefront pipeline, the per-pixel kernel and the literal true-division walk are
    three more GPU implementations: all must give the oracle's frame AND make exactly the reference's
    number of box and primitive tests (SURVEY.md 8(d) prices the roofline on those counts)."""
    r = request.getfixturevalue("renderer_" + which)
    sf = scene_util.load(name, 240, 160)
    P = sf.params
    if P.mode == rt.RT_MODE_PATHTRACE:
        P.mode = rt.RT_MODE_PHONG
    for bvh in (0, 1):
        desc = sf.desc
        desc.use_bvh = bvh
        assert_frame_parity(*run_both(r, sf, desc=desc, params=P), check_counts=True)


@pytest.mark.parametrize("name", ["clutter_500", "mirror_glass_cylinder", "binary_primitives", "textured"])
@pytest.mark.parametrize("which", ["grid", "block", "fused_whole"])
def test_selectable_pipelines_agree(request, which, name):
    """RT_PIPELINE=grid / block / fused with RT_SPLIT_ROOTS=0: the same frame, hit IDs and ray count as the
    oracle (and therefore as the default fw_roots_kernel + fw_kernel path), including the first-frame
    arena overflow and retry that 300x200 clutter_500 provokes."""
    r = request.getfixturevalue("renderer_" + which)
    sf = scene_util.load(name, 300, 200)
    assert_frame_parity(*run_both(r, sf))


@pytest.mark.parametrize("bvh", [0, 1])
def test_clutter_500_both_tie_rules(renderer, bvh, golden_hits):
    sf = scene_util.load("clutter_500", 300, 200)
    desc = sf.desc
    desc.use_bvh = bvh
    out = run_both(renderer, sf, desc=desc)
    assert_frame_parity(*out)
    assert np.array_equal(out[1], golden_hits[f"clutter_500:bvh{bvh}"].astype(np.int32))


@pytest.mark.parametrize("name,tonemap", [("binary_primitives", None), ("blinn_phong", None),
                                          ("mirror_glass_cylinder", None), ("clutter_500", None),
                                          ("mirror_glass_cylinder", "reinhard"), ("mirror_glass_cylinder", "ward"),
                                          ("mirror_glass_cylinder", "uncharted2"), ("textured", None)])
def test_full_size_ppm_against_reference_hash(renderer, name, tonemap, ref_outputs, tmp_path):
    """1200x800 as shipped.  The quantized image must be the reference's: a pow() last-bit difference
    can only flip a byte when 255.999*c sits within 1e-13 of an integer, so at most a handful of bytes
    may differ; in practice the file is byte-identical and the sha256 matches."""
    sf = scene_util.load(name, tonemap=tonemap)
    desc = sf.desc
    if name == "clutter_500":
        desc.use_bvh = 1  # same image either way; the linear scan is covered at 300x200
    rgb, hit, st, orgb, ohit, cnt = run_both(renderer, sf, desc=desc)
    assert_frame_parity(rgb, hit, st, orgb, ohit, cnt)
    key = name + (":" + tonemap if tonemap else "")
    if "rays" in ref_outputs[key]:
        assert st.rays == ref_outputs[key]["rays"]
    q_gpu = (255.999 * rgb).astype(np.int32)
    q_cpu = (255.999 * orgb).astype(np.int32)
    n_bytes_differ = int((q_gpu != q_cpu).sum())
    assert n_bytes_differ <= 4
    if n_bytes_differ == 0:
        p = os.path.join(str(tmp_path), "o.ppm")
        rt.write_ppm(p, rgb)
        assert hashlib.sha256(open(p, "rb").read()).hexdigest() == ref_outputs[key]["sha256"]


def test_clutter_500_4k_full_config(renderer):
    """BASELINE config 3 at its full size: 3840x2160, BVH tie rule, all rays."""
    sf = scene_util.load("clutter_500", 3840, 2160)
    desc = sf.desc
    desc.use_bvh = 1
    out = run_both(renderer, sf, desc=desc)
    assert_frame_parity(*out)
    assert out[2].rays > 30_000_000


@pytest.mark.parametrize("name", ["blinn_phong", "mirror_glass_cylinder"])
def test_1080p_config(renderer, name):
    sf = scene_util.load(name, 1920, 1080)
    assert_frame_parity(*run_both(renderer, sf))


@pytest.mark.parametrize("w,h", [(2, 2), (9, 7), (64, 8), (65, 9), (301, 203)])
def test_ragged_image_sizes(renderer, w, h):
    sf = scene_util.load("mirror_glass_cylinder", w, h)
    assert_frame_parity(*run_both(renderer, sf))


@pytest.mark.parametrize("depth", [0, 1, 2, 3])
def test_small_max_depth(renderer, depth):
    sf = scene_util.load("mirror_glass_cylinder", 150, 100)
    P = sf.params
    P.max_depth = depth
    rgb, hit, st, orgb, ohit, cnt = run_both(renderer, sf, params=P)
    if depth == 0:  # traceRay returns the background before intersecting: no rays, no hit IDs
        assert st.rays == 0 and cnt["rays"] == 0
        assert np.array_equal(rgb, orgb)
    else:
        assert_frame_parity(rgb, hit, st, orgb, ohit, cnt)


def test_empty_scene_and_no_lights(renderer):
    cam = scene_util.look_at((0, 0, -5), (0, 0, 0))

    class Holder:
        camera = cam
    for use_bvh in (0, 1):
        s = scene_util.SyntheticScene(np.zeros(0, dtype=np.int32), np.zeros((0, 12)), use_bvh=use_bvh)
        P = scene_util.make_params(40, 30)
        renderer.upload(s.desc); renderer.set_camera(cam); renderer.set_params(P)
        rgb, hit = renderer.render()
        assert (hit == -1).all() and np.allclose(rgb, [0.1, 0.2, 0.3], rtol=0, atol=0)
    rng = np.random.default_rng(3)
    types, geom = scene_util.random_prims(rng, 10, 10, 5)
    s = scene_util.SyntheticScene(types, geom)  # no lights: ambient only
    P = scene_util.make_params(60, 40)
    renderer.upload(s.desc); renderer.set_camera(cam); renderer.set_params(P)
    rgb, hit = renderer.render()
    orgb, ohit, _, _ = oracle_lib.render(s.desc, cam, P)
    assert np.array_equal(hit, ohit) and np.abs(rgb - orgb).max() <= TIGHT_ABS


@pytest.mark.parametrize("seed", [0, 1, 2])
def test_random_soup_with_mirrors_and_glass(renderer, seed):
    """Synthetic scenes: arbitrary-axis cylinders (wrong +Y boxes), overlapping primitives, three lights,
    mirror and glass materials; both tie rules."""
    rng = np.random.default_rng(100 + seed)
    types, geom = scene_util.random_prims(rng, 30, 30, 12)
    mats = [scene_util.default_material(),
            scene_util.default_material(diffuse=(0.2, 0.3, 0.9), reflective=1, reflectivity=0.6),
            scene_util.default_material(diffuse=(0.9, 0.9, 0.9), refractive=1, ior=1.5),
            scene_util.default_material(diffuse=(0.5, 0.9, 0.4), reflective=1, reflectivity=0.3, refractive=1, ior=1.33)]
    pm = rng.integers(0, len(mats), len(types)).astype(np.int32)
    lights = [scene_util.point_light((0, 8, -3)), scene_util.point_light((5, 2, 5), (0.5, 0.4, 0.3)),
              scene_util.area_light((0, 5, 0), (0, -1, 0), (1, 0, 0), (0, 0, 1), 1, 1)]
    cam = scene_util.look_at((0, 1, -9), (0, 0, 0))
    for use_bvh in (0, 1):
        s = scene_util.SyntheticScene(types, geom, materials=mats, lights=lights, prim_material=pm, use_bvh=use_bvh)
        P = scene_util.make_params(120, 80, max_depth=6)
        renderer.upload(s.desc); renderer.set_camera(cam); renderer.set_params(P)
        rgb, hit = renderer.render()
        st = renderer.stats()
        orgb, ohit, cnt, _ = oracle_lib.render(s.desc, cam, P)
        assert_frame_parity(rgb, hit, st, orgb, ohit, cnt)


def large_soup(seed=7, n_sph=1500, n_tri=300, n_cyl=100):
    """1900 primitives: the staged part of the scene image (filter nodes, leaf records, geometry) is
    several times the 150 KB that fit in LDS, so every kernel reads the image from HBM."""
    rng = np.random.default_rng(seed)
    types, geom = scene_util.random_prims(rng, n_sph, n_tri, n_cyl, extent=10.0)
    mats = [scene_util.default_material(),
            scene_util.default_material(diffuse=(0.2, 0.3, 0.9), reflective=1, reflectivity=0.6),
            scene_util.default_material(diffuse=(0.9, 0.9, 0.9), refractive=1, ior=1.5),
            scene_util.default_material(diffuse=(0.5, 0.9, 0.4), reflective=1, reflectivity=0.3, refractive=1, ior=1.33)]
    pm = rng.integers(0, len(mats), len(types)).astype(np.int32)
    lights = [scene_util.point_light((0, 25, -10)), scene_util.point_light((15, 6, 15), (0.5, 0.4, 0.3)),
              scene_util.area_light((0, 14, 0), (0, -1, 0), (1, 0, 0), (0, 0, 1), 2, 2)]
    cam = scene_util.look_at((0, 3, -32), (0, 0, 0))
    return types, geom, mats, pm, lights, cam


@pytest.mark.parametrize("which", ["renderer", "renderer_block", "renderer_grid", "renderer_levels", "renderer_mega"])
def test_scene_too_large_for_lds(request, which):
    """The non-staged form of every pipeline (scene image in HBM), both tie rules."""
    r = request.getfixturevalue(which)
    types, geom, mats, pm, lights, cam = large_soup()
    for use_bvh in ((0, 1) if which == "renderer" else (1,)):
        s = scene_util.SyntheticScene(types, geom, materials=mats, lights=lights, prim_material=pm, use_bvh=use_bvh)
        P = scene_util.make_params(96, 64, max_depth=5)
        r.upload(s.desc); r.set_camera(cam); r.set_params(P)
        rgb, hit = r.render()
        st = r.stats()
        assert st.scene_in_lds == 0
        orgb, ohit, cnt, _ = oracle_lib.render(s.desc, cam, P)
        assert_frame_parity(rgb, hit, st, orgb, ohit, cnt)


def test_culling_never_changes_a_hit_on_loose_cylinder_boxes(renderer, renderer_walk):
    """Cylinders with arbitrary axes stick out of the boxes the reference gives them
    (src/Cylinder.cpp:131-137); nodes above them must never be culled by entry distance."""
    rng = np.random.default_rng(77)
    types, geom = scene_util.random_prims(rng, 10, 10, 60)
    o = rng.uniform(-7, 7, (300_000, 3))
    d = rng.standard_normal((300_000, 3))
    s = scene_util.SyntheticScene(types, geom, use_bvh=1)
    renderer.upload(s.desc)
    renderer_walk.upload(s.desc)
    a = renderer.trace_rays(o, d)
    b = renderer_walk.trace_rays(o, d)
    ot, oprim, onrm = oracle_lib.trace_rays(s.desc, o, d)
    for got in (a, b):
        assert np.array_equal(got[1], oprim) and np.array_equal(got[0], ot) and np.array_equal(got[2], onrm)


def test_band_partition_reassembles_bit_exactly(renderer):
    """Multi-GPU split of a Whitted frame: bands of 8 rows dealt round-robin; each rank's compact
    output must be exactly the corresponding rows of the one-GPU frame."""
    sf = scene_util.load("mirror_glass_cylinder", 200, 150)  # 150 rows: last band is ragged
    renderer.load(sf)
    full, full_hit = renderer.render()
    for world in (2, 3, 8):
        got = np.zeros_like(full)
        got_hit = np.zeros_like(full_hit)
        for rank in range(world):
            P = sf.params
            P.band_rank, P.band_world = rank, world
            renderer.set_params(P)
            part, part_hit = renderer.render()
            rows = [j for j in range(150) if (j // 8) % world == rank]
            assert part.shape[0] == len(rows)
            got[rows] = part
            got_hit[rows] = part_hit
        assert np.array_equal(got, full) and np.array_equal(got_hit, full_hit)


def test_render_twice_is_deterministic(renderer):
    sf = scene_util.load("clutter_500", 400, 300)
    renderer.load(sf)
    a, ha = renderer.render()
    b, hb = renderer.render()
    assert np.array_equal(a, b) and np.array_equal(ha, hb)


def test_bad_arguments_are_rejected(renderer):
    sf = scene_util.load("blinn_phong", 64, 64)
    P = sf.params
    P.width = 1
    with pytest.raises(rt.RtError):
        renderer.set_params(P)
    P = sf.params
    P.max_depth = 1000
    with pytest.raises(rt.RtError):
        renderer.set_params(P)
    P = sf.params
    P.band_rank, P.band_world = 3, 2
    with pytest.raises(rt.RtError):
        renderer.set_params(P)
    s = scene_util.SyntheticScene(np.array([7], dtype=np.int32), np.zeros((1, 12)))
    with pytest.raises(rt.RtError):
        renderer.upload(s.desc)
    renderer.load(sf)  # leave the shared context usable


def test_quantize_device_is_the_ppm_conversion(renderer):
    """rt_quantize_device == int(255.999 * c) (RayTracer.cpp:359-362) on a rendered frame, device to device."""
    import torch
    sf = scene_util.load("mirror_glass_cylinder", 300, 200)
    renderer.load(sf)
    n = 300 * 200 * 3
    rgb = torch.empty(n, dtype=torch.float64, device="cuda:0")
    u8 = torch.empty(n, dtype=torch.uint8, device="cuda:0")
    s = torch.cuda.current_stream().cuda_stream
    renderer.render_device(rgb.data_ptr(), None, stream=s)
    renderer.quantize_device(rgb.data_ptr(), n, u8.data_ptr(), stream=s)
    torch.cuda.synchronize()
    host = rgb.cpu().numpy()
    assert np.array_equal(u8.cpu().numpy(), (255.999 * host).astype(np.int32).astype(np.uint8))
    orgb, _, _, _ = oracle_lib.render(sf.desc, sf.camera, sf.params, want_hit=False)
    assert np.abs(host.reshape(200, 300, 3) - orgb).max() <= TIGHT_ABS


def test_workgroup_cap_and_frames_in_flight(renderer):
    """rt_set_workgroups: the frame does not depend on the grid size, and several contexts rendering at
    once on their own streams (bench.py's frames in flight) each produce the one-context frame."""
    import torch
    sf = scene_util.load("clutter_500", 300, 200)
    desc = sf.desc; desc.use_bvh = 1
    renderer.upload(desc); renderer.set_camera(sf.camera); renderer.set_params(sf.params)
    ref, ref_hit = renderer.render()
    rays = renderer.stats().rays
    ctxs, streams, outs = [], [], []
    try:
        for wg in (1, 7, 40):
            r = rt.Renderer(0)
            r.upload(desc); r.set_camera(sf.camera); r.set_params(sf.params)
            r.set_workgroups(wg)
            ctxs.append(r)
            streams.append(torch.cuda.Stream())
            outs.append((torch.zeros(200 * 300 * 3, dtype=torch.float64, device="cuda"),
                         torch.full((200 * 300,), -2, dtype=torch.int32, device="cuda")))
        for _ in range(3):  # the first round sizes each context's node store, the others overlap
            for r, s, (rgb, hit) in zip(ctxs, streams, outs):
                r.render_device(rgb.data_ptr(), hit.data_ptr(), stream=s.cuda_stream)
        torch.cuda.synchronize()
        for r, (rgb, hit) in zip(ctxs, outs):
            assert np.array_equal(rgb.cpu().numpy().reshape(ref.shape), ref)
            assert np.array_equal(hit.cpu().numpy().reshape(ref_hit.shape), ref_hit)
            assert r.stats().rays == rays
        with pytest.raises(rt.RtError):
            ctxs[0].set_workgroups(-1)
    finally:
        for r in ctxs:
            r.close()


@pytest.mark.parametrize("which", ["renderer", "renderer_block", "renderer_grid", "renderer_levels", "renderer_mega"])
def test_fused_ppm_bytes_equal_the_separate_conversion(request, which):
    """rt_render_device_u8 == rt_render_device + rt_quantize_device, with and without the fp64 frame."""
    import torch
    r = request.getfixturevalue(which)
    for name in ("clutter_500", "binary_primitives"):
        sf = scene_util.load(name, 301, 203)
        r.load(sf, sf.params)
        ref, ref_hit = r.render()
        n = ref.size
        expect = (255.999 * ref).astype(np.int32).astype(np.uint8).reshape(-1)
        u8 = torch.zeros(n, dtype=torch.uint8, device="cuda")
        rgb = torch.zeros(n, dtype=torch.float64, device="cuda")
        hit = torch.full((n // 3,), -2, dtype=torch.int32, device="cuda")
        r.render_device_u8(u8.data_ptr(), None, None)
        torch.cuda.synchronize()
        assert np.array_equal(u8.cpu().numpy(), expect)
        u8.zero_()
        r.render_device_u8(u8.data_ptr(), rgb.data_ptr(), hit.data_ptr())
        torch.cuda.synchronize()
        assert np.array_equal(u8.cpu().numpy(), expect)
        assert np.array_equal(rgb.cpu().numpy().reshape(ref.shape), ref)
        assert np.array_equal(hit.cpu().numpy().reshape(ref_hit.shape), ref_hit)
    sf = scene_util.load("pathtrace", 32, 24)
    P = sf.params; P.mode = rt.RT_MODE_PATHTRACE
    r.load(sf, P)
    with pytest.raises(rt.RtError):
        r.render_device_u8(u8.data_ptr(), None, None)


@pytest.mark.parametrize("n_spheres,bvh", [(5000, True), (5000, False), (60000, True)])
def test_generated_clutter_scenes(renderer, tmp_path, n_spheres, bvh):
    """seeded large clutter scenes (scripts/gen_clutter.py) -- the filter BVH's top is staged in
    LDS, the rest of the image is read from HBM; both tie rules."""
    import importlib.util
    import json
    spec = importlib.util.spec_from_file_location("gen_clutter", os.path.join(ROOT, "scripts", "gen_clutter.py"))
    gen = importlib.util.module_from_spec(spec)
    spec.loader.exec_module(gen)
    path = tmp_path / "clutter.json"
    path.write_text(json.dumps(gen.clutter(n_spheres, seed=3, width=192, height=128, bvh=bvh)))
    sf = rt.SceneFile(str(path))
    renderer.load(sf, sf.params)
    rgb, hit = renderer.render()
    st = renderer.stats()
    assert st.scene_in_lds == 0
    orgb, ohit, cnt, _ = oracle_lib.render(sf.desc, sf.camera, sf.params)
    assert_frame_parity(rgb, hit, st, orgb, ohit, cnt)


@pytest.mark.parametrize("name,tonemap", [("binary_primitives", None), ("mirror_glass_cylinder", None), ("clutter_500", None)])
def test_cli_writes_the_reference_file(name, tonemap, ref_outputs, tmp_path):
    """The drop-in binary, invoked as the reference is (`raytracer scene.json out.ppm`): its file has the
    reference's sha256; it renders PPM bytes on the device (rt_render_u8 / rt_write_ppm_u8)."""
    import subprocess
    exe = os.path.join(ROOT, "raytracer_amd", "csrc", "raytracer")
    out = tmp_path / "o.ppm"
    res = subprocess.run([exe, os.path.join("scenes", name + ".json"), str(out)] + ([tonemap] if tonemap else []),
                         cwd=ROOT, capture_output=True, text=True, timeout=120)
    assert res.returncode == 0, res.stderr
    assert "Rendering complete. Image saved to" in res.stdout
    assert hashlib.sha256(out.read_bytes()).hexdigest() == ref_outputs[name]["sha256"]


def test_cli_environment_overrides(renderer, tmp_path):
    """RT_WIDTH / RT_HEIGHT, RT_PIXEL_SAMPLES / RT_LIGHT_SAMPLES and RT_FIX_SAMPLE_SWAP leave the
    original invocation alone and give the file the library renders with those parameters."""
    import subprocess
    exe = os.path.join(ROOT, "raytracer_amd", "csrc", "raytracer")

    def cli(scene, env):
        out = tmp_path / "o.ppm"
        res = subprocess.run([exe, os.path.join("scenes", scene + ".json"), str(out)], cwd=ROOT, capture_output=True, text=True,
                             timeout=120, env=dict(os.environ, **env))
        assert res.returncode == 0, res.stderr
        return out.read_bytes(), res.stdout

    def lib(sf, P):
        renderer.load(sf, P)
        u8, _ = renderer.render_u8()
        ref = tmp_path / "ref.ppm"
        rt.write_ppm_u8(ref, u8)
        return ref.read_bytes()

    sf = scene_util.load("blinn_phong", 300, 200)
    got, _ = cli("blinn_phong", {"RT_WIDTH": "300", "RT_HEIGHT": "200"})
    assert got == lib(sf, sf.params)
    # path trace: pathtrace.json has neither sample key -> pixelSamples 4, lightSamples 16 after the reference's swap
    sf = scene_util.load("pathtrace", 60, 40)
    small = {"RT_WIDTH": "60", "RT_HEIGHT": "40"}
    got, out = cli("pathtrace", small)
    assert "Pixel samples: 4" in out and "Light samples: 16" in out
    assert got == lib(sf, sf.params)
    got, out = cli("pathtrace", dict(small, RT_FIX_SAMPLE_SWAP="1"))
    assert "Pixel samples: 16" in out and "Light samples: 4" in out
    P = sf.params
    P.pixel_samples, P.light_samples = 16, 4
    assert got == lib(sf, P)
    got, out = cli("pathtrace", dict(small, RT_PIXEL_SAMPLES="9", RT_LIGHT_SAMPLES="2", RT_SEED="7"))
    assert "Pixel samples: 9" in out and "Light samples: 2" in out
    P.pixel_samples, P.light_samples, P.seed = 9, 2, 7
    assert got == lib(sf, P)


def test_production_library_has_the_traversal_counters_compiled_out(renderer):
    """The filter-BVH traversal's test counters cost 2-6 %; they belong to variants/libraytracer_amd_counts.so only
    (a Makefile rule once built the production objects with -DRT_FBVH_COUNTS=1 unnoticed)."""
    sf = scene_util.load("clutter_500", 150, 100)
    renderer.load(sf)
    renderer.render()
    st = renderer.stats()
    assert st.rays > 0 and st.box_tests == 0 and st.prim_tests == 0


def test_render_u8_is_the_quantized_frame(renderer, tmp_path):
    """rt_render_u8 / rt_write_ppm_u8 == rt_render + rt_write_ppm, Whitted and path-traced."""
    for name, mode in (("clutter_500", None), ("pathtrace", rt.RT_MODE_PATHTRACE)):
        sf = scene_util.load(name, 150, 100)
        P = sf.params
        if mode is not None:
            P.mode, P.pixel_samples, P.light_samples = mode, 4, 2
        renderer.load(sf, P)
        rgb, hit = renderer.render()
        u8, hit8 = renderer.render_u8(want_hit=True)
        assert np.array_equal(u8, (255.999 * rgb).astype(np.int32).astype(np.uint8))
        assert np.array_equal(hit8, hit)
        a, b = tmp_path / "a.ppm", tmp_path / "b.ppm"
        rt.write_ppm(a, rgb); rt.write_ppm_u8(b, u8)
        assert a.read_bytes() == b.read_bytes()


def test_frame_pipeline_with_frames_in_flight_on_streams(renderer):
    """bench.py's loop at world size 1: 4 contexts of 16 workgroups on 4 streams that were measured to overlap
    (dist.concurrent_streams), PPM bytes from the kernels, 10 frames dealt round-robin through FramePipeline."""
    import torch
    from raytracer_amd import dist as rtd
    sf = scene_util.load("mirror_glass_cylinder", 304, 200)
    renderer.load(sf, sf.params)
    ref, _ = renderer.render()
    expect = (255.999 * ref).astype(np.int32).astype(np.uint8)
    dev = torch.device("cuda", 0)
    try:
        streams = rtd.concurrent_streams(4, dev)
        assert rtd.concurrent_streams(2, dev) == streams[:2]  # cached
    except RuntimeError:  # fewer hardware queues than on the MI355X boxes: overlap is a matter of speed only
        streams = [torch.cuda.Stream(device=dev) for _ in range(4)]
    assert len(streams) == 4 and len({s.cuda_stream for s in streams}) == 4
    ctxs = []
    try:
        for _ in range(4):
            r = rt.Renderer(0)
            r.load(sf, sf.params)
            r.set_workgroups(16)
            ctxs.append(r)
        H, W = sf.params.height, sf.params.width
        pipe = rtd.FramePipeline(0, 1, H, W, dev, band_rows=sf.params.band_rows, depth=4, streams=streams)
        frames = []
        n = 10
        def finish(k):
            with torch.cuda.stream(pipe.stream(k)):
                frames.append(pipe.complete(k).clone())
        for k in range(n):
            with torch.cuda.stream(pipe.stream(k)):
                _, u8 = pipe.target(k)
                u8.zero_()
                ctxs[k % 4].render_device_u8(u8.data_ptr(), None, None, stream=torch.cuda.current_stream().cuda_stream)
                pipe.submit(k)
            if k >= 3:
                finish(k - 3)
        for k in range(n - 3, n):
            finish(k)
        torch.cuda.synchronize()
        assert len(frames) == n
        for f in frames:
            assert np.array_equal(f.cpu().numpy(), expect)
    finally:
        for r in ctxs:
            r.close()


def test_counting_traversal_walks_every_ray(renderer_walk, renderer):
    """A ray whose answer cannot reach the image (the shadow rays of a hit that refracts: RayTracer.cpp:798 replaces its
    local colour) is counted but not walked by the production traversal; the walk traversal, which makes the reference's
    own test counts, starts a traversal for every ray it counts, and both give the same frame."""
    sf = scene_util.load("mirror_glass_cylinder", 300, 200)
    out = []
    for r in (renderer_walk, renderer):
        r.load(sf)
        rgb, hit = r.render()
        out.append((rgb, hit, r.stats()))
    assert out[0][2].rays_traversed == out[0][2].rays == out[1][2].rays
    assert out[1][2].rays_traversed == 0  # (compiled out of the production library, like its test counters)
    assert np.array_equal(out[0][1], out[1][1]) and np.array_equal(out[0][0], out[1][0])
