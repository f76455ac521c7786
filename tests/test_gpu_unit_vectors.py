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
