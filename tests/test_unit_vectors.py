"""The oracle against per-function known answers recorded FROM THE REFERENCE'S OWN OBJECTS
(tests/golden/unit_vectors.npz, generated in the build container by tests/golden/make_unit_vectors.py: our driver
linked with the reference's translation units).  This pins the oracle to the reference on the edge cases the GPU
is compared to the oracle on: slab tests with zero direction components and origins on the planes
(src/BoundingBox.cpp:24-62), primitive tests on grazing / inside / axis-perpendicular rays
(src/Sphere.cpp:14-48, src/Triangle.cpp:42-80, src/Cylinder.cpp:13-107), Scene::intersect with both tie rules,
refract / fresnel / fresnelReflectance / reflect incl. total internal reflection (src/RayTracer.cpp:430-504),
toneMap (:161-202).  Bit-exact: same compiler family, same libm, same operation order."""
import numpy as np
import pytest

from tests import oracle_lib, unit_vector_cases as uv


@pytest.fixture(scope="module")
def V():
    return uv.vectors()


def test_fixture_covers_the_edge_cases(V):
    b = V["box_in"]
    assert (b[:, 9:12] == 0).any(axis=1).sum() > 1000            # zero direction components
    on_plane = ((b[:, 6:9] == b[:, 0:3]) | (b[:, 6:9] == b[:, 3:6])).any(axis=1)
    assert (on_plane & (b[:, 9:12] == 0).any(axis=1)).sum() > 300  # 0/0 candidates
    assert 0.1 < V["box_out"].mean() < 0.9
    o = V["optics_out"]
    assert (o[:, 3] == 0).sum() > 200 and (o[:, 3] == 1).sum() > 2000   # total internal reflection and transmission
    for name in ("sphere", "triangle", "cylinder_y", "cylinder_x", "cylinder_d"):
        assert 0.05 < V[name + "_out"][:, 0].mean() < 0.95, name


def test_box_slab_test_is_the_references(V):
    b, want = V["box_in"], V["box_out"][:, 0] > 0
    got = np.array([oracle_lib.box_hit(r[0:6], r[6:9], r[9:12]) for r in b])
    assert np.array_equal(got, want), f"{(got != want).sum()} of {len(want)} box tests differ"


@pytest.mark.parametrize("name", ["sphere", "triangle", "cylinder_y", "cylinder_x", "cylinder_d"])
def test_primitive_intersections_are_the_references(V, name):
    rays = V[name + "_in"]
    t, prim, nrm = oracle_lib.trace_rays(uv.one_prim_scene(name).desc, rays[:, 0:3], rays[:, 3:6])
    uv.check_primitive(name, V, t, prim, nrm)


@pytest.mark.parametrize("key,bvh", [("scene_linear", 0), ("scene_bvh", 1)])
def test_scene_intersect_is_the_references(V, key, bvh):
    rays = V[key + "_in"]
    t, prim, nrm = oracle_lib.trace_rays(uv.five_prim_scene(bvh).desc, rays[:, 0:3], rays[:, 3:6])
    uv.check_scene(key, V, t, prim, nrm)


def test_optics_are_the_references(V):
    got, want = oracle_lib.optics(V["optics_in"]), V["optics_out"]
    assert np.array_equal(got[:, :8], want[:, :8], equal_nan=True)
    # fresnelReflectance: pow(x, 5.0) of the same libm
    assert np.array_equal(got[:, 8], want[:, 8])


def test_tone_maps_are_the_references(V):
    assert np.array_equal(oracle_lib.tonemap(V["tonemap_in"]), V["tonemap_out"], equal_nan=True)
