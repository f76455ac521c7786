"""Known-answer tests of the oracle's building blocks and of the host-side BVH flattener."""
import ctypes as C

import numpy as np
import pytest

from raytracer_amd import capi
from tests import oracle_lib, scene_util


def philox(ctr, key):
    c = np.array(ctr, dtype=np.uint32)
    k = np.array(key, dtype=np.uint32)
    o = np.zeros(4, dtype=np.uint32)
    oracle_lib.load().oracle_philox(c.ctypes.data, k.ctypes.data, o.ctypes.data)
    return [int(x) for x in o]


def test_philox_known_answers():
    # Random123 kat_vectors, philox4x32-10
    assert philox([0, 0, 0, 0], [0, 0]) == [0x6627e8d5, 0xe169c58d, 0xbc57ac4c, 0x9b00dbd8]
    assert philox([0xffffffff] * 4, [0xffffffff] * 2) == [0x408f276d, 0x41c83b0e, 0xa20bc7c6, 0x6d5451fd]
    assert philox([0x243f6a88, 0x85a308d3, 0x13198a2e, 0x03707344], [0xa4093822, 0x299f31d0]) == \
        [0xd16cfe09, 0x94fdcceb, 0x5001e420, 0x24126ea1]


def test_uniform_stream_is_53_bit_and_keyed():
    lib = oracle_lib.load()
    o = philox([7, 3, 5, 1], [11, 0])
    expect_even = ((o[1] << 32 | o[0]) >> 11) * 2.0 ** -53
    expect_odd = ((o[3] << 32 | o[2]) >> 11) * 2.0 ** -53
    assert lib.oracle_uniform(11, 7, 3, 5, 2) == expect_even
    assert lib.oracle_uniform(11, 7, 3, 5, 3) == expect_odd
    u = np.array([lib.oracle_uniform(0, p, 0, 1, 0) for p in range(4000)])
    assert 0.0 <= u.min() and u.max() < 1.0 and abs(u.mean() - 0.5) < 0.02


def box_hit(box, o, d):
    b, oo, dd = (np.array(x, dtype=np.float64) for x in (box, o, d))
    return bool(oracle_lib.load().oracle_box_hit(b.ctypes.data, oo.ctypes.data, dd.ctypes.data))


def test_box_slab_cases():
    unit = [0, 0, 0, 1, 1, 1]
    assert box_hit(unit, [0.5, 0.5, -1], [0, 0, 1])          # zero components: +-inf slabs
    assert not box_hit(unit, [2.0, 0.5, -1], [0, 0, 1])
    assert not box_hit(unit, [0.5, 0.5, 2], [0, 0, 1])       # behind: tmax <= 0
    assert box_hit(unit, [0.5, 0.5, 0.5], [0, 0, 1])         # inside
    # origin ON a slab plane with a zero direction component: 0/0 = NaN and every compare with it is
    # false (src/BoundingBox.cpp:25-61).  On the MIN plane tmin = NaN is later overwritten and the
    # box is hit; on the MAX plane tmax = NaN survives to `return tmax > 0` and the box is missed.
    assert box_hit(unit, [0.0, 0.5, -1], [0, 0, 1])
    assert not box_hit(unit, [1.0, 0.5, -1], [0, 0, 1])
    assert box_hit([0, 0, 0, 1, 1, 0], [0.25, 0.25, -1], [0, 0, 1])  # flat box (axis-aligned triangle)


@pytest.mark.parametrize("tm", [0, 1, 2, 3])
def test_tonemap_curves(tm):
    lib = oracle_lib.load()
    c = np.array([0.25, 1.0, 4.0])
    out = np.zeros(3)
    lib.oracle_tonemap(c.ctypes.data, tm, out.ctypes.data)
    if tm == 0:
        expect = c
    elif tm == 1:
        expect = c / (c + 1)
    elif tm == 2:
        r = c / (c + 1)
        expect = r * (r * (r * 0.4 + 0.4) + 1.219) / (r * (r * 0.3 + 0.5) + 2.219)
    else:
        f = np.float32
        A, B, Cc, D, E, F = f(0.15), f(0.50), f(0.10), f(0.20), f(0.02), f(0.30)

        def part(x):
            return ((x * (x * float(A) + float(Cc * B)) + float(D * E)) / (x * (x * float(A) + float(B)) + float(D * F))) - float(E / F)
        expect = part(c * 2.0) * (1.0 / part(np.array([float(f(11.2))] * 3)))
    assert np.array_equal(out, expect)


def test_flattened_bvh_matches_oracle_tree():
    """rt_build_bvh (product, host) and the oracle's pointer tree are built independently; leaf order
    and node count must agree on scenes with many ties (Cornell box) and on clutter_500."""
    lib = capi.load()
    for name in ["clutter_500", "final", "mirror_glass_cylinder", "binary_primitives"]:
        sf = scene_util.load(name)
        desc = sf.desc
        desc.use_bvh = 1
        n = lib.rt_build_bvh(C.byref(desc), 0, None, None, None)
        box = np.zeros((n, 6)); left = np.zeros(n, dtype=np.int32); right = np.zeros(n, dtype=np.int32)
        assert lib.rt_build_bvh(C.byref(desc), n, box.ctypes.data, left.ctypes.data, right.ctypes.data) == n
        order = []

        def walk(i):
            for ch in (left[i], right[i]):
                if ch == -2 ** 31:
                    continue
                if ch < 0:
                    order.append(~int(ch))
                else:
                    walk(int(ch))
        walk(0)
        o_order = np.zeros(desc.n_prims, dtype=np.int32)
        o_nodes = np.zeros(1, dtype=np.int32)
        k = oracle_lib.load().oracle_bvh_leaf_order(desc, o_order.ctypes.data, o_nodes.ctypes.data)
        assert k == desc.n_prims and int(o_nodes[0]) == n
        assert order == [int(x) for x in o_order]
        assert sorted(order) == list(range(desc.n_prims))


def test_random_rays_linear_vs_bvh_agree_in_oracle_for_plus_y_cylinders():
    # with conservative boxes (spheres, triangles, +Y cylinders) the BVH only prunes, so t must agree
    rng = np.random.default_rng(5)
    types, geom = scene_util.random_prims(rng, 40, 40, 10, axis_aligned_cyl=True)
    o = rng.uniform(-6, 6, (2000, 3)); d = rng.standard_normal((2000, 3))
    s1 = scene_util.SyntheticScene(types, geom, use_bvh=1)
    s0 = scene_util.SyntheticScene(types, geom, use_bvh=0)
    t1, p1, _ = oracle_lib.trace_rays(s1.desc, o, d)
    t0, p0, _ = oracle_lib.trace_rays(s0.desc, o, d)
    assert np.array_equal(t1, t0)
    assert (p1 >= 0).sum() > 100


def test_texture_row_on_a_cylinders_top_cap_hangs_on_the_points_last_bit():
    """Cylinder getUV: v = clamp(y / height, 0, 1) (src/Cylinder.cpp); Material::getTextureColor wraps v - floor(v) and takes
    row int((1 - v) h) (src/Material.cpp:69-86).  On the TOP cap y / height is 1 or 1 - 2^-53: the first wraps to 0 and lands
    in the texture's LAST row, the second in its FIRST.  Which one a hit gets depends on the last bit of its point -- the same
    on the GPU and in the oracle for the same point (Whitted frames are exact), not for the points of a path tracer after a
    diffuse bounce, whose cos / sin differ in the last bits between libraries (scripts/gpu_fuzz.py keeps textures off
    cylinders in path-traced cases for this reason)."""
    tex = np.zeros((4, 1, 3), dtype=np.uint8)
    tex[:, 0, 0] = [10, 20, 30, 40]  # row k has red = 10 (k + 1)
    geom = np.zeros((1, capi.RT_GEOM_STRIDE))
    geom[0, 0:3] = (0.3, -0.7, 1.1)
    geom[0, 3:6] = (0.0, 1.0, 0.0)
    geom[0, 6], geom[0, 7] = 1.0, 1.3
    mat = scene_util.default_material()
    mat.texture = 0
    syn = scene_util.SyntheticScene(np.array([capi.RT_CYLINDER], dtype=np.int32), geom, materials=[mat], textures=[tex])
    top = geom[0, 1] + geom[0, 7]
    rng = np.random.default_rng(2)
    n = 2000
    pts = np.column_stack([0.3 + rng.uniform(-0.5, 0.5, n), np.full(n, top), 1.1 + rng.uniform(-0.5, 0.5, n)])
    pts[:, 1] = np.nextafter(pts[:, 1], np.where(rng.random(n) < 0.5, -np.inf, np.inf))  # one ulp below / above the cap's plane
    out = oracle_lib.uv_texture(syn.desc, np.zeros(n, dtype=np.int32), points=pts)
    rows = np.rint(out[:, 2] * 255.0 / 10.0).astype(int) - 1
    assert set(rows.tolist()) == {0, 3}            # never a row in between: the cap's v is 1 up to rounding ...
    assert 0.2 < (rows == 0).mean() < 0.8          # ... and which end of the texture it means is a toss of the last bit
