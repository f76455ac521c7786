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


# ---- the deterministic half of the path tracer (src/RayTracer.cpp:506-637), getUV, getTextureColor -----------------------

def test_path_fixture_covers_the_branches(V):
    out = V["pt_trace_bvh_out"]
    assert np.isfinite(out).all() and (out >= 0).all()
    bg = V["pt_setup"][0][17:20]
    sky = (out == bg).all(axis=1)
    assert 0.05 < sky.mean() < 0.9                               # rays that leave at once, and rays that do not
    assert np.array_equal(V["pt_trace_bvh_out"], V["pt_trace_linear_out"])  # (no equal-t ties in this scene: both tie rules agree)
    d = V["pt_direct_bvh_out"]
    assert (d[:, 0] > 0).mean() > 0.3 and (d[d[:, 0] > 0, 1:] > 0).any(axis=1).mean() > 0.3  # lit and shadowed hits


@pytest.mark.parametrize("bvh", [0, 1])
def test_trace_ray_path_is_the_references(V, bvh):
    """traceRayPath(ray, 0) on render()'s camera rays over mirrors, glass (both branches traced, total internal reflection
    through the NaN of :572) and a point light, maxDepth 4: bit for bit."""
    scene, cam, P = uv.pt_scene(V, bvh)
    _, _, cnt, acc = oracle_lib.render(scene.desc, cam, P, want_hit=False, want_accum=True)
    want = V["pt_trace_bvh_out" if bvh else "pt_trace_linear_out"].reshape(P.height, P.width, 3)
    assert np.array_equal(acc, want)
    assert cnt["rays"] > 3 * P.width * P.height


@pytest.mark.parametrize("bvh", [0, 1])
def test_estimate_direct_light_point_branch_is_the_references(V, bvh):
    """estimateDirectLight at the primary hits, view = -ray.direction.normalize(): the (e + 2) / (2 pi) normalisation, the
    unflipped normal, shadow test t < distance (:602-637)."""
    scene, cam, P = uv.pt_scene(V, bvh)
    o, d = uv.camera_rays(cam, P.width, P.height)
    got = oracle_lib.direct_light(scene.desc, P, o, d)
    want = V["pt_direct_bvh_out" if bvh else "pt_direct_linear_out"]
    assert np.array_equal(got[:, 0], want[:, 0])
    assert np.array_equal(got[:, 1:], want[:, 1:])


@pytest.mark.parametrize("k,name", list(enumerate(uv.SCENE_ORDER)))
def test_get_uv_is_the_references(V, k, name):
    rows = V["uv_" + name]
    got = oracle_lib.uv_texture(uv.textured_five_prim_scene().desc, np.full(len(rows), k), points=rows[:, 0:3])
    assert np.array_equal(got[:, 0:2], rows[:, 3:5])   # same libm: atan2 / acos bit for bit


def test_get_texture_color_is_the_references(V):
    rows = V["texture_checkerboard"]
    scene = uv.textured_five_prim_scene()
    assert [scene.texels[0].shape[1], scene.texels[0].shape[0]] == list(V["texture_checkerboard_size"][0])
    got = oracle_lib.uv_texture(scene.desc, np.zeros(len(rows), dtype=np.int32), uv=rows[:, 0:2])
    assert np.array_equal(got[:, 2:5], rows[:, 2:5])
    assert len(np.unique(rows[:, 2:5], axis=0)) >= 2     # (both colours of the board)
