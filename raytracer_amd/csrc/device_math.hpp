// device_math.hpp -- fp64 vector arithmetic and primitive tests for gfx950, written so that every
// result has the bits the reference's x86-64 build produces: same operation order as
// src/Vector3.cpp, true IEEE divisions and square roots, no FMA contraction (the translation
// unit is compiled with -ffp-contract=off), explicit compares instead of min/max intrinsics so
// NaN cases fall the way the reference's do.
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>

#define RT_DEV __device__ __forceinline__

namespace rtd {

struct D3 {
    double x, y, z;
};
RT_DEV D3 mk(double x, double y, double z) { return D3{x, y, z}; }
RT_DEV D3 operator+(D3 a, D3 b) { return D3{a.x + b.x, a.y + b.y, a.z + b.z}; }
RT_DEV D3 operator-(D3 a, D3 b) { return D3{a.x - b.x, a.y - b.y, a.z - b.z}; }
RT_DEV D3 operator*(D3 a, D3 b) { return D3{a.x * b.x, a.y * b.y, a.z * b.z}; }
RT_DEV D3 operator/(D3 a, D3 b) { return D3{a.x / b.x, a.y / b.y, a.z / b.z}; }
RT_DEV D3 operator*(D3 a, double s) { return D3{a.x * s, a.y * s, a.z * s}; }
RT_DEV D3 operator/(D3 a, double s) { return D3{a.x / s, a.y / s, a.z / s}; }
RT_DEV D3 operator+(D3 a, double s) { return D3{a.x + s, a.y + s, a.z + s}; }
RT_DEV D3 operator-(D3 a, double s) { return D3{a.x - s, a.y - s, a.z - s}; }
RT_DEV D3 operator-(D3 a) { return D3{-a.x, -a.y, -a.z}; }
RT_DEV double dot(D3 a, D3 b) { return a.x * b.x + a.y * b.y + a.z * b.z; }  // Vector3.cpp:48-50
RT_DEV D3 cross(D3 a, D3 b) { return D3{a.y * b.z - a.z * b.y, a.z * b.x - a.x * b.z, a.x * b.y - a.y * b.x}; }
// sqrt() of the device library is rsq, a Goldschmidt step and two correction steps, wrapped in a rescaling for
// arguments below 2^-767 and a pass-through for 0 / inf (6 of its 17 instructions).  For arguments in
// [2^-766, 2^1022) this is the same sequence without the wrapping, hence the same (correctly rounded) bits; others
// take the library call.  tests/test_gpu_rays.py::test_division_and_sqrt_are_correctly_rounded covers it.
RT_DEV double sqrt_windowed(double x) {
    if ((unsigned)__double2hiint(x) - 0x10100000u >= 0x6FD00000u) return sqrt(x);  // also negative, NaN
    double y = __builtin_amdgcn_rsq(x);
    double g = x * y, h = y * 0.5;
    double r = __builtin_fma(-h, g, 0.5);
    g = __builtin_fma(g, r, g);
    h = __builtin_fma(h, r, h);
    double d = __builtin_fma(-g, g, x);
    g = __builtin_fma(d, h, g);
    d = __builtin_fma(-g, g, x);
    return __builtin_fma(d, h, g);
}
// sqrt(x) and 1 / sqrt(x) from ONE reciprocal-square-root estimate: the library sequence's Goldschmidt step refines
// both, one correction step on the root (within an ulp of sqrt(x); the reciprocal within two).  For callers that are
// held to a tolerance, not to the reference's bits (pt_kernel's regrouped light sample).  x = 0, negative or NaN
// gives NaNs, which those callers' guards turn into a zero contribution.
RT_DEV void sqrt_and_rsqrt(double x, double& root, double& inv) {
    const double y = __builtin_amdgcn_rsq(x);
    double g = x * y, h = y * 0.5;
    const double r = __builtin_fma(-h, g, 0.5);
    g = __builtin_fma(g, r, g);
    h = __builtin_fma(h, r, h);
    const double d = __builtin_fma(-g, g, x);
    root = __builtin_fma(d, h, g);
    inv = h + h;
}
RT_DEV double length(D3 a) { return sqrt_windowed(a.x * a.x + a.y * a.y + a.z * a.z); }  // Vector3.cpp:72-75
RT_DEV double refine_rcp(double d);
RT_DEV double div_shared(double a, double d, double r);
// Vector3.cpp:78-81: three true divisions by the same length.  hipcc expands each into 11-12 instructions; when
// no quotient needs v_div_scale's scaling -- every component in [2^-400, inf) in magnitude and the length at most
// 2^400, so that quotients lie in [2^-800, 1] -- the same bits come from one refined reciprocal and 3
// instructions per component (see refine_rcp / div_shared below).  Zero, tiny, huge or non-finite components
// take the literal divisions.  tests/test_gpu_rays.py::test_normalize_shares_one_reciprocal_exactly.
RT_DEV D3 normalize(D3 a) {
    const double len = length(a);
    const unsigned hx = (unsigned)__double2hiint(a.x) & 0x7FFFFFFFu, hy = (unsigned)__double2hiint(a.y) & 0x7FFFFFFFu,
                   hz = (unsigned)__double2hiint(a.z) & 0x7FFFFFFFu;
    const unsigned smallest = min(min(hx, hy), hz);
    if (smallest >= 0x26F00000u && (unsigned)__double2hiint(len) <= 0x58F00000u) {  // 2^-400, 2^400
        const double r = refine_rcp(len);
        return D3{div_shared(a.x, len, r), div_shared(a.y, len, r), div_shared(a.z, len, r)};
    }
    return D3{a.x / len, a.y / len, a.z / len};
}
RT_DEV D3 ld3(const double* p) { return D3{p[0], p[1], p[2]}; }

// std::max / std::min / std::clamp with the libstdc++ argument order (NaN handling)
RT_DEV double std_max(double a, double b) { return (a < b) ? b : a; }
RT_DEV double std_min(double a, double b) { return (b < a) ? b : a; }
RT_DEV double std_clamp(double v, double lo, double hi) { return (v < lo) ? lo : ((hi < v) ? hi : v); }

// BoundingBox::intersect (src/BoundingBox.cpp:24-62); b = min xyz, max xyz.
RT_DEV bool box_hit(const double* b, D3 o, D3 d) {
    double tmin = (b[0] - o.x) / d.x;
    double tmax = (b[3] - o.x) / d.x;
    if (tmin > tmax) { double s = tmin; tmin = tmax; tmax = s; }
    double tymin = (b[1] - o.y) / d.y;
    double tymax = (b[4] - o.y) / d.y;
    if (tymin > tymax) { double s = tymin; tymin = tymax; tymax = s; }
    if ((tmin > tymax) || (tymin > tmax)) return false;
    if (tymin > tmin) tmin = tymin;
    if (tymax < tmax) tmax = tymax;
    double tzmin = (b[2] - o.z) / d.z;
    double tzmax = (b[5] - o.z) / d.z;
    if (tzmin > tzmax) { double s = tzmin; tzmin = tzmax; tzmax = s; }
    if ((tmin > tzmax) || (tzmin > tmax)) return false;
    if (tzmax < tmax) tmax = tzmax;
    return tmax > 0;
}

// Sphere::intersect (src/Sphere.cpp:14-36), t only; s = cx cy cz r.
RT_DEV bool sphere_t(const double* s, D3 o, D3 d, double& t_out) {
    D3 oc = o - ld3(s);
    double a = dot(d, d);
    double b = 2.0 * dot(oc, d);
    double c = dot(oc, oc) - s[3] * s[3];
    double disc = b * b - 4 * a * c;
    if (disc < 0) return false;
    double sq = sqrt(disc);
    double t0 = (-b - sq) / (2.0 * a);
    double t1 = (-b + sq) / (2.0 * a);
    double t = t0;
    if (t < 0) {
        t = t1;
        if (t < 0) return false;
    }
    t_out = t;
    return true;
}

// ---- exact division with a shared reciprocal ---------------------------------------------------
// hipcc expands an f64 a/d into  v_div_scale(d), v_div_scale(a), v_rcp, 2 Newton steps (4 fma),
// q = a*r, rem = fma(-d, q, a), v_div_fmas(rem, r, q), v_div_fixup  -- 11 instructions, of which the
// first 6 depend on d only.  When neither operand needs scaling (see ray_regular below) the two
// v_div_scale are identities, v_div_fmas is a plain fma and v_div_fixup passes its first operand
// through, so the SAME bits come out of: refine_rcp(d) once, then 3 instructions per quotient.
// A ray divides dozens of box extents by the same three direction components, which is where
// BoundingBox::intersect's time goes (6 true divisions per node, BoundingBox.cpp:25-45).
// tests/test_gpu_rays.py::test_shared_reciprocal_division_is_exact checks the identity on the GPU.
RT_DEV double refine_rcp(double d) {
    double r = __builtin_amdgcn_rcp(d);
    double e = __builtin_fma(-d, r, 1.0);
    r = __builtin_fma(r, e, r);
    e = __builtin_fma(-d, r, 1.0);
    r = __builtin_fma(r, e, r);
    return r;
}
RT_DEV double div_shared(double a, double d, double r) {
    double q = a * r;
    double rem = __builtin_fma(-d, q, a);
    return __builtin_fma(rem, r, q);
}
// a / d for three numerators over one denominator (D3 / double of the reference: AreaLight.cpp:13, RayTracer.cpp:678),
// bit-identical to the three true divisions: when |d| and every |a_k| lie in [2^-300, 2^300] no operand or quotient
// can need v_div_scale's scaling, and one refined reciprocal serves all three (11 + 2 x 3 instructions instead of
// 3 x 11); zero, tiny, huge or non-finite operands take the literal divisions.
RT_DEV D3 div3(D3 a, double d) {
    const unsigned hx = (unsigned)__double2hiint(a.x) & 0x7FFFFFFFu, hy = (unsigned)__double2hiint(a.y) & 0x7FFFFFFFu,
                   hz = (unsigned)__double2hiint(a.z) & 0x7FFFFFFFu, hd = (unsigned)__double2hiint(d) & 0x7FFFFFFFu;
    const unsigned lo = min(min(hx, hy), min(hz, hd)), hi = max(max(hx, hy), max(hz, hd));
    if (lo >= 0x2D300000u && hi <= 0x52B00000u) {  // 2^-300, 2^300
        const double r = refine_rcp(d);
        return D3{div_shared(a.x, d, r), div_shared(a.y, d, r), div_shared(a.z, d, r)};
    }
    return D3{a.x / d, a.y / d, a.z / d};
}

// |x| == 0 or 2^-300 <= |x| <= 2^300 (false for NaN / inf)
RT_DEV bool in_window(double x) {
    double ax = fabs(x);
    return ax == 0.0 || (ax >= 0x1p-300 && ax <= 0x1p300);
}
// A ray for which every slab quotient (box - o)/d stays clear of v_div_scale's scaling cases and
// of NaN: direction components in [2^-150, 2], origin components 0 or in [2^-300, 2^300].  With box
// coordinates in the same window (checked on upload) numerators are 0 or >= 2^-353 in magnitude
// and quotients lie in [2^-354, 2^451].
RT_DEV bool ray_regular(D3 o, D3 d) {
    double ax = fabs(d.x), ay = fabs(d.y), az = fabs(d.z);
    bool dir_ok = ax >= 0x1p-150 && ax <= 2.0 && ay >= 0x1p-150 && ay <= 2.0 && az >= 0x1p-150 && az <= 2.0;
    return dir_ok && in_window(o.x) && in_window(o.y) && in_window(o.z);
}

// BoundingBox::intersect for a regular ray.  No NaN can occur, so the reference's swap / early-out
// sequence reduces exactly to  enter = max of the three slab minima, exit = min of the three slab
// maxima, hit <=> enter <= exit && exit > 0  (min/max are exact operations).
RT_DEV bool box_hit_regular(const double* b, D3 o, D3 d, D3 r, double& t_enter) {
    double x0 = div_shared(b[0] - o.x, d.x, r.x), x1 = div_shared(b[3] - o.x, d.x, r.x);
    double y0 = div_shared(b[1] - o.y, d.y, r.y), y1 = div_shared(b[4] - o.y, d.y, r.y);
    double z0 = div_shared(b[2] - o.z, d.z, r.z), z1 = div_shared(b[5] - o.z, d.z, r.z);
    double enter = fmax(fmax(fmin(x0, x1), fmin(y0, y1)), fmin(z0, z1));
    double exit = fmin(fmin(fmax(x0, x1), fmax(y0, y1)), fmax(z0, z1));
    t_enter = enter;
    return (enter <= exit) && (exit > 0);
}

// numerator n of a quotient n / (2a) that may use the shared reciprocal of 2a ~ 2
RT_DEV bool num_ok(double n) {
    double an = fabs(n);
    return an == 0.0 || (an >= 0x1p-900 && an <= 0x1p900);
}

// Sphere::intersect with a = d.d, 2a and refine_rcp(2a) hoisted out (same values every call).
RT_DEV bool sphere_t_shared(const double* s, D3 o, D3 d, double a, double two_a, double r2a, double& t_out) {
    D3 oc = o - ld3(s);
    double b = 2.0 * dot(oc, d);
    double c = dot(oc, oc) - s[3] * s[3];
    double disc = b * b - 4 * a * c;
    if (disc < 0) return false;
    double sq = sqrt(disc);
    double n0 = -b - sq, n1 = -b + sq;
    double t0, t1;
    if (num_ok(n0) && num_ok(n1)) {
        t0 = div_shared(n0, two_a, r2a);
        t1 = div_shared(n1, two_a, r2a);
    } else {
        t0 = n0 / two_a;
        t1 = n1 / two_a;
    }
    double t = t0;
    if (t < 0) {
        t = t1;
        if (t < 0) return false;
    }
    t_out = t;
    return true;
}

// Triangle::intersect (src/Triangle.cpp:42-80), t only; tr = v0, e1 = v1-v0, e2 = v2-v0, n.
RT_DEV bool triangle_t(const double* tr, D3 o, D3 d, double& t_out) {
    const double EPS = 1e-8;
    D3 v0 = ld3(tr), e1 = ld3(tr + 3), e2 = ld3(tr + 6);
    D3 h = cross(d, e2);
    double a = dot(e1, h);
    if (fabs(a) < EPS) return false;
    double f = 1.0 / a;
    D3 s = o - v0;
    double u = f * dot(s, h);
    if (u < 0.0 || u > 1.0) return false;
    D3 q = cross(s, e1);
    double v = f * dot(d, q);
    if (v < 0.0 || u + v > 1.0) return false;
    double t = f * dot(e2, q);
    if (t > EPS) {
        t_out = t;
        return true;
    }
    return false;
}

// Cylinder::intersect (src/Cylinder.cpp:13-107); cy = base, axis, r, h.
// part: 0 side, 1 bottom cap, 2 top cap (selects the normal afterwards).
RT_DEV bool cylinder_t(const double* cy, D3 o, D3 dir, double& t_out, int& part_out) {
    D3 base = ld3(cy), axis = ld3(cy + 3);
    double radius = cy[6], height = cy[7];
    D3 oc = o - base;
    D3 d = dir - axis * dot(dir, axis);
    D3 ocp = oc - axis * dot(oc, axis);
    double a = dot(d, d);
    double b = 2.0 * dot(d, ocp);
    double c = dot(ocp, ocp) - radius * radius;
    double disc = b * b - 4 * a * c;
    double t = __builtin_inf();
    int part = 0;
    bool hit = false;
    if (disc >= 0) {
        double sq = sqrt(disc);
        double t0 = (-b - sq) / (2 * a);
        double t1 = (-b + sq) / (2 * a);
        if (t0 > t1) { double s = t0; t0 = t1; t1 = s; }
        double y0 = dot(o + dir * t0 - base, axis);
        if (t0 >= 0 && y0 >= 0 && y0 <= height) {
            t = t0;
            hit = true;
        } else {
            double y1 = dot(o + dir * t1 - base, axis);
            if (t1 >= 0 && y1 >= 0 && y1 <= height) {
                t = t1;
                hit = true;
            }
        }
    }
    double tb = dot(base - o, axis) / dot(dir, axis);
    if (tb >= 0) {
        D3 p = o + dir * tb;
        D3 dd = p - base;
        if (dot(dd, dd) <= radius * radius) {
            if (tb < t) {
                t = tb;
                part = 1;
                hit = true;
            }
        }
    }
    D3 top = base + axis * height;
    double tt = dot(top - o, axis) / dot(dir, axis);
    if (tt >= 0) {
        D3 p = o + dir * tt;
        D3 dd = p - top;
        if (dot(dd, dd) <= radius * radius) {
            if (tt < t) {
                t = tt;
                part = 2;
                hit = true;
            }
        }
    }
    t_out = t;
    part_out = part;
    return hit;
}

// One value of a lane's row of 16 lanes to a partner lane, as a data-parallel-primitive move (no LDS round trip):
// kCtrl 0xB1 / 0x4E = quad_perm [1,0,3,2] / [2,3,0,1] (lane ^ 1, lane ^ 2), 0x141 = row_half_mirror (7 - lane within
// 8 lanes), 0x140 = row_mirror (15 - lane within the row).
template <int kCtrl>
RT_DEV double dpp_f64(double v) {
    // (every lane has a source lane under these controls: no "old" value, which would cost a v_mov per half)
    const int lo = __builtin_amdgcn_mov_dpp(__double2loint(v), kCtrl, 0xF, 0xF, true);
    const int hi = __builtin_amdgcn_mov_dpp(__double2hiint(v), kCtrl, 0xF, 0xF, true);
    return __hiloint2double(hi, lo);
}

// Philox4x32-10 and the uniform stream of DESIGN.md "RNG" (same definition as oracle/oracle.cpp,
// written independently; both are pinned to the Random123 known-answer vectors).
RT_DEV void philox4x32_10(uint32_t c0, uint32_t c1, uint32_t c2, uint32_t c3, uint32_t k0, uint32_t k1,
                          uint32_t out[4]) {
#pragma unroll
    for (int r = 0; r < 10; ++r) {
        // one full 32 x 32 -> 64 product each (v_mad_u64_u32) instead of a mul_hi / mul_lo pair: integer
        // multiplies run at a quarter of the VALU rate and are two thirds of this function's time
        const uint64_t p0 = (uint64_t)0xD2511F53u * c0, p1 = (uint64_t)0xCD9E8D57u * c2;
        uint32_t hi0 = (uint32_t)(p0 >> 32), lo0 = (uint32_t)p0;
        uint32_t hi1 = (uint32_t)(p1 >> 32), lo1 = (uint32_t)p1;
        uint32_t n0 = hi1 ^ c1 ^ k0;
        uint32_t n2 = hi0 ^ c3 ^ k1;
        c0 = n0; c1 = lo1; c2 = n2; c3 = lo0;
        k0 += 0x9E3779B9u;
        k1 += 0xBB67AE85u;
    }
    out[0] = c0; out[1] = c1; out[2] = c2; out[3] = c3;
}

struct Rng {
    uint32_t k0, k1, pixel, sample;
};
RT_DEV double rng_u(const Rng& r, uint32_t node, uint32_t dim) {
    uint32_t o[4];
    philox4x32_10(r.pixel, r.sample, node, dim >> 1, r.k0, r.k1, o);
    uint64_t bits = (dim & 1) ? (((uint64_t)o[3] << 32) | o[2]) : (((uint64_t)o[1] << 32) | o[0]);
    return (double)(bits >> 11) * 0x1.0p-53;
}
// the two uniforms of one counter block: rng_u(r, node, 2 * block) and rng_u(r, node, 2 * block + 1) from ONE Philox
// call (written as two rng_u calls the compiler cannot see that dim >> 1 is the same block when dim is computed)
RT_DEV void rng_u2(const Rng& r, uint32_t node, uint32_t block, double& u0, double& u1) {
    uint32_t o[4];
    philox4x32_10(r.pixel, r.sample, node, block, r.k0, r.k1, o);
    u0 = (double)((((uint64_t)o[1] << 32) | o[0]) >> 11) * 0x1.0p-53;
    u1 = (double)((((uint64_t)o[3] << 32) | o[2]) >> 11) * 0x1.0p-53;
}

}  // namespace rtd
