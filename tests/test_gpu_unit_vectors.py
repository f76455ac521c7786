"""The HIP path against the known answers recorded from the reference's own objects
(tests/golden/unit_vectors.npz, see tests/test_unit_vectors.py): Scene::intersect through rt_debug_trace_rays with
every traversal, the optics and tone-map device functions through rt_debug_optics.  Bit-exact, except
fresnelReflectance, whose pow(x, 5.0) the device evaluates as x^2 * x^2 * x (<= 2 ulp)."""
import numpy as np
import pytest

from tests import unit_vector_cases as uv

pytestmark = pytest.mark.gpu


@pytest.fixture(scope="module")
def V():
    return uv.vectors()


@pytest.mark.parametrize("which", ["renderer", "renderer_walk", "renderer_literal"])
@pytest.mark.parametrize("name", ["sphere", "triangle", "cylinder_y", "cylinder_x", "cylinder_d"])
def test_primitive_intersections_are_the_references(request, V, which, name):
    r = request.getfixturevalue(which)
    rays = V[name + "_in"]
    r.upload(uv.one_prim_scene(name).desc)
    t, prim, nrm = r.trace_rays(rays[:, 0:3], rays[:, 3:6])
    uv.check_primitive(name, V, t, prim, nrm)


@pytest.mark.parametrize("which", ["renderer", "renderer_walk", "renderer_literal"])
@pytest.mark.parametrize("key,bvh", [("scene_linear", 0), ("scene_bvh", 1)])
def test_scene_intersect_is_the_references(request, V, which, key, bvh):
    """Zero direction components, origins on the gate boxes' planes (0/0 = NaN), the x-axis cylinder whose reference
    box assumes +Y, both tie rules."""
    r = request.getfixturevalue(which)
    rays = V[key + "_in"]
    r.upload(uv.five_prim_scene(bvh).desc)
    t, prim, nrm = r.trace_rays(rays[:, 0:3], rays[:, 3:6])
    uv.check_scene(key, V, t, prim, nrm)


def test_optics_are_the_references(renderer, V):
    got, want = renderer.optics(0, V["optics_in"]), V["optics_out"]
    assert np.array_equal(got[:, :8], want[:, :8], equal_nan=True)   # refract, TIR test, reflect, fresnel
    assert np.abs(got[:, 8] - want[:, 8]).max() <= 4 * np.finfo(float).eps  # fresnelReflectance (values <= 1)


def test_tone_maps_are_the_references(renderer, V):
    assert np.array_equal(renderer.optics(1, V["tonemap_in"]), V["tonemap_out"], equal_nan=True)


# ---- the deterministic half of the path tracer, getUV, getTextureColor ---------------------------------------------------

@pytest.mark.parametrize("queues", ["1", "0"])
@pytest.mark.parametrize("bvh", [0, 1])
def test_trace_ray_path_is_the_references(V, queues, bvh):
    """traceRayPath on render()'s camera rays (pixelSamples 1, RT_DEBUG_NO_JITTER) over mirrors, glass and a point light,
    through both production kernels: pq_kernel (queues = 1; sums T x term exactly, rounds once) and pt_kernel (0; combines
    bottom-up like the reference).  The specular power is square-and-multiply on the device: 1e-13."""
    import torch
    from tests.conftest import _make
    scene, cam, P = uv.pt_scene(V, bvh)
    want = V["pt_trace_bvh_out" if bvh else "pt_trace_linear_out"]
    r = _make({"RT_PT_QUEUES": queues})
    try:
        r.upload(scene.desc); r.set_camera(cam); r.set_params(P)
        acc = torch.empty(P.width * P.height * 3, dtype=torch.float64, device="cuda")
        r.accumulate_device(acc.data_ptr(), stream=torch.cuda.current_stream().cuda_stream)
        torch.cuda.synchronize()
        got = acc.cpu().numpy().reshape(-1, 3)
        assert np.abs(got - want).max() <= 1e-13 * max(1.0, want.max())
        sky = (want == V["pt_setup"][0][17:20]).all(axis=1)
        assert np.array_equal(got[sky], want[sky])      # a ray that leaves at once returns the background's bits
    finally:
        r.close()


@pytest.mark.parametrize("k,name", list(enumerate(uv.SCENE_ORDER)))
def test_get_uv_is_the_references(renderer, V, k, name):
    """atan2 / acos of the device library differ from glibc's in the last bits (<= 2 ulp)."""
    rows = V["uv_" + name]
    renderer.upload(uv.textured_five_prim_scene().desc)
    got = renderer.optics(2, np.column_stack([np.full(len(rows), float(k)), rows[:, 0:3]]))
    assert np.abs(got[:, 0:2] - rows[:, 3:5]).max() <= 1e-15


def test_get_texture_color_is_the_references(renderer, V):
    rows = V["texture_checkerboard"]
    renderer.upload(uv.textured_five_prim_scene().desc)
    got = renderer.optics(3, np.column_stack([np.zeros(len(rows)), rows[:, 0:2], np.zeros(len(rows))]))
    assert np.array_equal(got[:, 2:5], rows[:, 2:5])   # wrap, truncation, clamp and byte / 255.0: no transcendental involved
