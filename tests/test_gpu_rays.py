"""Scene::intersect parity ray by ray (rt_debug_trace_rays vs the oracle) and device math exactness."""
import numpy as np
import pytest

from tests import oracle_lib, scene_util

pytestmark = pytest.mark.gpu


def compare_rays(renderer, desc, o, d):
    renderer.upload(desc)
    t, prim, nrm = renderer.trace_rays(o, d)
    ot, oprim, onrm = oracle_lib.trace_rays(desc, o, d)
    assert np.array_equal(prim, oprim), f"{(prim != oprim).sum()} of {len(prim)} primitive IDs differ"
    assert np.array_equal(t, ot)          # bit-exact t (division and sqrt are correctly rounded)
    assert np.array_equal(nrm, onrm)
    return prim


def test_division_and_sqrt_are_correctly_rounded(renderer):
    rng = np.random.default_rng(11)
    n = 4_000_000
    a = rng.standard_normal(n) * np.exp(rng.uniform(-200, 200, n))
    b = rng.standard_normal(n) * np.exp(rng.uniform(-200, 200, n))
    a[:8] = [0.0, -0.0, 1.0, np.inf, -np.inf, 0.0, 5e-324, 1e308]
    b[:8] = [0.0, 1.0, 0.0, 2.0, -0.0, -3.0, 3.0, 1e-308]
    with np.errstate(all="ignore"):
        expect = a / b
    got = renderer.math(0, a, b)
    assert np.array_equal(got, expect, equal_nan=True)
    assert np.array_equal(np.signbit(got), np.signbit(expect))
    x = np.abs(a)
    assert np.array_equal(renderer.math(1, x), np.sqrt(x))


def test_windowed_sqrt_is_the_library_sqrt(renderer):
    """device_math.hpp sqrt_windowed (op 13, behind every length()): the device library's sequence without its
    rescaling wrapper inside [2^-766, 2^1022), the library call outside -- correctly rounded either way, window edges,
    perfect squares, mantissas next to 1 and 4, zeros, denormals, inf and NaN included."""
    rng = np.random.default_rng(17)
    n = 6_000_000
    with np.errstate(over="ignore"):
        x = np.exp2(rng.uniform(-1074, 1024, n)) * rng.uniform(1, 2, n)  # a few overflow to inf: also a case
    m = 200_000
    near = np.frombuffer((np.uint64(0x3FF0000000000000) + rng.integers(-6, 7, m).astype(np.int64).astype(np.uint64)).tobytes(),
                         dtype=np.float64)
    x[:m] = near * np.exp2(rng.integers(-800, 1023, m).astype(np.float64))          # just around powers of two
    k = rng.integers(1, 2 ** 26, m).astype(np.float64)
    x[m:2 * m] = k * k * np.exp2(2.0 * rng.integers(-300, 300, m))                  # perfect squares
    edge = np.array([0x10100000, 0x100FFFFF, 0x7FE00000, 0x7FDFFFFF, 0x7FEFFFFF, 0x00100000, 0x000FFFFF], dtype=np.uint64)
    x[2 * m:2 * m + 7] = np.frombuffer(((edge << np.uint64(32)) | np.uint64(0xFFFFFFFF)).tobytes(), dtype=np.float64)
    x[2 * m + 7:2 * m + 14] = np.frombuffer((edge << np.uint64(32)).tobytes(), dtype=np.float64)
    x[2 * m + 14:2 * m + 20] = [0.0, -0.0, np.inf, np.nan, 5e-324, -1.0]
    with np.errstate(all="ignore"):
        expect = np.sqrt(x)
    got = renderer.math(13, x)
    assert np.array_equal(got, expect, equal_nan=True)
    assert np.array_equal(np.signbit(got[~np.isnan(expect)]), np.signbit(expect[~np.isnan(expect)]))


def test_shared_reciprocal_division_is_exact(renderer):
    """device_math.hpp div_shared: op 8 evaluates a / d through refine_rcp(d) + 3 instructions; inside
    the window ray_regular() guarantees it must equal the IEEE quotient bit for bit, including
    mantissas of all ones and numerators of 0."""
    rng = np.random.default_rng(13)
    n = 6_000_000
    a = rng.standard_normal(n) * np.exp2(rng.uniform(-350, 300, n))
    d = np.copysign(np.exp2(rng.uniform(-150, 1, n)), rng.standard_normal(n))
    d = np.clip(np.abs(d), 2.0 ** -150, 2.0) * np.sign(d)
    # adversarial mantissas
    m = 200_000
    ones = np.frombuffer((np.uint64(0x3FEFFFFFFFFFFFFF) - rng.integers(0, 4, m).astype(np.uint64)).tobytes(), dtype=np.float64)
    d[:m] = ones * np.exp2(rng.integers(-149, 1, m).astype(np.float64))
    a[m:2 * m] = np.frombuffer((np.uint64(0x3FF0000000000000) + rng.integers(0, 4, m).astype(np.uint64)).tobytes(), dtype=np.float64)
    a[2 * m:2 * m + 1000] = 0.0
    got = renderer.math(8, a, d)
    assert np.array_equal(got, a / d)


def test_normalize_shares_one_reciprocal_exactly(renderer):
    """device_math.hpp normalize: one refined reciprocal of the length when no quotient can need v_div_scale's
    scaling, the literal divisions otherwise -- either way the bits of Vector3::normalize (x / sqrt(x*x + y*y + z*z))."""
    rng = np.random.default_rng(17)
    n = 4_000_000
    x = rng.standard_normal(n) * np.exp2(rng.uniform(-30, 30, n))
    y = rng.standard_normal(n) * np.exp2(rng.uniform(-30, 30, n))
    # around the guard's limits, far outside them, zeros, adversarial mantissas
    k = 400_000
    x[:k] = rng.standard_normal(k) * np.exp2(rng.uniform(-450, 450, k))
    y[:k] = rng.standard_normal(k) * np.exp2(rng.uniform(-450, 450, k))
    x[k:k + 1000] = 0.0
    y[k + 500:k + 1500] = 0.0
    x[2 * k:3 * k] = np.frombuffer((np.uint64(0x3FEFFFFFFFFFFFFF) - rng.integers(0, 4, k).astype(np.uint64)).tobytes(), dtype=np.float64)
    y[3 * k:4 * k] = np.frombuffer((np.uint64(0x3FF0000000000000) + rng.integers(0, 4, k).astype(np.uint64)).tobytes(), dtype=np.float64)
    x[-4:] = [np.inf, 1e-320, 1e300, -0.0]
    y[-4:] = [1.0, 1e-320, 1e300, 1.0]
    z = 0.75 * y
    with np.errstate(all="ignore"):
        ex = x / np.sqrt(x * x + y * y + z * z)
        ez = x / np.sqrt(y * y + z * z + x * x)
    got = renderer.math(9, x, y)
    assert np.array_equal(got, ex, equal_nan=True)
    assert np.array_equal(np.signbit(got), np.signbit(ex))
    got = renderer.math(10, x, y)
    assert np.array_equal(got, ez, equal_nan=True)
    assert np.array_equal(np.signbit(got), np.signbit(ez))


def test_vector_over_scalar_shares_one_reciprocal_exactly(renderer):
    """device_math.hpp div3 (Vector3 / double: AreaLight.cpp:13, RayTracer.cpp:678): one refined reciprocal for the
    three quotients when no operand can need v_div_scale's scaling, the literal divisions otherwise -- either way
    the correctly rounded IEEE quotients."""
    rng = np.random.default_rng(23)
    n = 3_000_000
    x = rng.standard_normal(n) * np.exp2(rng.uniform(-40, 40, n))
    d = rng.standard_normal(n) * np.exp2(rng.uniform(-40, 40, n))
    k = 300_000
    x[:k] = rng.standard_normal(k) * np.exp2(rng.uniform(-350, 350, k))   # around and beyond the guard's limits
    d[:k] = rng.standard_normal(k) * np.exp2(rng.uniform(-350, 350, k))
    x[k:k + 1000] = 0.0
    d[k + 500:k + 1500] = 0.0
    x[2 * k:3 * k] = np.frombuffer((np.uint64(0x3FEFFFFFFFFFFFFF) - rng.integers(0, 4, k).astype(np.uint64)).tobytes(), dtype=np.float64)
    d[3 * k:4 * k] = np.frombuffer((np.uint64(0x3FF0000000000000) + rng.integers(0, 4, k).astype(np.uint64)).tobytes(), dtype=np.float64)
    x[-5:] = [np.inf, 1e-320, 1e300, -0.0, np.nan]
    d[-5:] = [1.0, 1e-320, 1e-300, 1.0, 1.0]
    with np.errstate(all="ignore"):
        e = x / d
    for op in (11, 12):
        got = renderer.math(op, x, d)
        assert np.array_equal(got, e, equal_nan=True)
        finite = ~np.isnan(e)
        assert np.array_equal(np.signbit(got[finite]), np.signbit(e[finite]))


def test_transcendentals_within_a_few_ulp(renderer):
    """pow/atan2/acos/cos/sin of the device library are not bit-identical to glibc; the contract only
    needs colours within 1e-4.  Measured: <= 2 ulp."""
    rng = np.random.default_rng(12)
    n = 1_000_000
    base, ex = rng.uniform(0, 1, n), rng.integers(1, 200, n).astype(np.float64)
    for op, a, b, ref in [(2, base, ex, np.power(base, ex)),
                          (3, rng.standard_normal(n), rng.standard_normal(n), None),
                          (4, rng.uniform(-1, 1, n), None, None),
                          (5, rng.uniform(0, 6.3, n), None, None), (6, rng.uniform(0, 6.3, n), None, None)]:
        if ref is None:
            ref = {3: lambda: np.arctan2(a, b), 4: lambda: np.arccos(a), 5: lambda: np.cos(a), 6: lambda: np.sin(a)}[op]()
        got = renderer.math(op, a, b)
        ulp = np.abs(got - ref) / np.maximum(np.spacing(np.abs(ref)), 5e-324)
        assert ulp.max() <= 4, (op, ulp.max())


def test_device_rng_equals_oracle_rng(renderer):
    lib = oracle_lib.load()
    pix = np.arange(0, 5000, dtype=np.float64)
    smp = (np.arange(0, 5000) % 37).astype(np.float64)
    got = renderer.math(7, pix, smp)
    expect = np.array([lib.oracle_uniform(0, int(p), int(s), 1, 0) for p, s in zip(pix, smp)])
    assert np.array_equal(got, expect)


@pytest.mark.parametrize("name,bvh", [("clutter_500", 1), ("clutter_500", 0), ("final", 1), ("mirror_glass_cylinder", 1),
                                      ("binary_primitives", 1), ("umbrella", 1)])
def test_random_rays_on_shipped_scenes(renderer, name, bvh):
    sf = scene_util.load(name)
    desc = sf.desc
    desc.use_bvh = bvh
    rng = np.random.default_rng(21)
    n = 200_000 if bvh else 20_000
    o = rng.uniform(-3, 3, (n, 3)) + np.array([0, 1, 0])
    d = rng.standard_normal((n, 3))
    prim = compare_rays(renderer, desc, o, d)
    assert (prim >= 0).mean() > 0.001  # binary_primitives is three small objects


@pytest.mark.parametrize("seed", range(4))
@pytest.mark.parametrize("bvh", [0, 1])
def test_random_rays_on_random_soups(renderer, seed, bvh):
    rng = np.random.default_rng(40 + seed)
    types, geom = scene_util.random_prims(rng, 60, 60, 25)
    s = scene_util.SyntheticScene(types, geom, use_bvh=bvh)
    n = 100_000
    o = rng.uniform(-6, 6, (n, 3))
    d = rng.standard_normal((n, 3))
    compare_rays(renderer, s.desc, o, d)


def test_degenerate_rays(renderer):
    """Axis-parallel rays (zero direction components -> +-inf and NaN slabs), origins on box planes,
    origins inside primitives, rays grazing spheres and running along triangle edges."""
    rng = np.random.default_rng(50)
    types, geom = scene_util.random_prims(rng, 20, 20, 10)
    # add axis-aligned geometry with exactly representable coordinates
    extra_t = [1, 1, 0, 2]
    extra_g = np.zeros((4, 12))
    extra_g[0, :9] = [0, 0, 2, 1, 0, 2, 0, 1, 2]          # triangle in the z = 2 plane
    extra_g[1, :9] = [-1, -0.5, -1, 1, -0.5, -1, 1, -0.5, 1]  # floor triangle
    extra_g[2, :4] = [0, 0, -3, 1]                        # unit sphere
    extra_g[3, :8] = [2, 0, 0, 0, 1, 0, 0.5, 1.0]         # +Y cylinder
    types = np.concatenate([types, np.array(extra_t, dtype=np.int32)])
    geom = np.concatenate([geom, extra_g])
    o_list, d_list = [], []
    axes = np.eye(3)
    for ax in range(3):
        for sgn in (1.0, -1.0):
            oo = rng.integers(-4, 5, (3000, 3)).astype(np.float64) * 0.5   # many land exactly on box planes
            o_list.append(oo)
            d_list.append(np.tile(axes[ax] * sgn, (3000, 1)))
    # rays in coordinate planes (one zero component)
    oo = rng.integers(-4, 5, (6000, 3)).astype(np.float64) * 0.25
    dd = rng.standard_normal((6000, 3))
    dd[np.arange(6000), rng.integers(0, 3, 6000)] = 0.0
    o_list.append(oo); d_list.append(dd)
    # grazing the unit sphere and along the triangle edge
    o_list.append(np.array([[1.0, 0.0, -10.0], [0.0, 1.0, -10.0], [0.0, 0.0, 2.0], [-1.0, 0.0, 2.0], [0.0, 0.0, -3.0]]))
    d_list.append(np.array([[0.0, 0.0, 1.0], [0.0, 0.0, 1.0], [1.0, 0.0, 0.0], [1.0, 0.0, 0.0], [0.3, 0.2, 0.1]]))
    o = np.concatenate(o_list); d = np.concatenate(d_list)
    for bvh in (0, 1):
        s = scene_util.SyntheticScene(types, geom, use_bvh=bvh)
        compare_rays(renderer, s.desc, o, d)
