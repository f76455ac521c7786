// rt_device.hpp -- device side shared by every translation unit of libraytracer_amd.so: scene view, closest-hit
// traversals, shading helpers (this file) and the kernel templates of the pipelines (the .inc files).  Everything
// lives in an anonymous namespace: each .hip instantiates the kernels it exports and nothing else, so the
// families compile in parallel (render_hip.hip: host side, per-pixel / level / grid kernels; k_blockwave.hip:
// bw_kernel; k_fused.hip: fw_roots_kernel, fw_kernel; k_pathwave.hip: pt_kernel; k_pathqueue.hip: pq_kernel).
#pragma once
#include <hip/hip_runtime.h>

#include <algorithm>
#include <chrono>
#include <cmath>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <initializer_list>
#include <map>
#include <tuple>
#include <vector>

#include "device_math.hpp"
#include "rt_internal.hpp"

#ifndef M_PI
#define M_PI 3.14159265358979323846
#endif

using namespace rtd;

// ------------------------------------------------------------------------------------------------
// Device-side scene description
// ------------------------------------------------------------------------------------------------
namespace {

constexpr int kNodeBytes = 64;  // 6 doubles box, int skip, int ref0, int ref1, int pad
constexpr int kLeafBytes = 64;   // filter BVH leaf record: primitive ref, gate flag, tie key, pad, gate box 6 doubles
constexpr int kFNodeBytes = 96;  // filter BVH: 6 floats box, leaf first, leaf count, 8 x (pass, fail) links as image offsets
constexpr int kSphereDoubles = 4;
constexpr int kTriDoubles = 12;
constexpr int kCylDoubles = 8;
constexpr int kBlock = 512;          // threads per workgroup (per-pixel kernel)
#ifndef RT_TRACE_BLOCK
#define RT_TRACE_BLOCK 1024
#endif
constexpr int kTraceBlock = RT_TRACE_BLOCK;  // threads per workgroup of the frame kernels: one LDS scene image per 16 waves
#ifndef RT_FUSED_BLOCK
#define RT_FUSED_BLOCK 768
#endif
#ifndef RT_PT_BLOCK
#define RT_PT_BLOCK RT_TRACE_BLOCK
#endif
#ifndef RT_PT_WAVES
#define RT_PT_WAVES 4
#endif
constexpr int kPtBlock = RT_PT_BLOCK;  // pt_kernel's own workgroup size; RT_PT_WAVES waves per SIMD (128 registers) whatever it is
#ifndef RT_FW_BLOCK
#define RT_FW_BLOCK RT_FUSED_BLOCK
#endif
constexpr int kFwBlock = RT_FW_BLOCK;  // fw_kernel's own workgroup size (fw_roots_kernel: kFusedBlock)
constexpr int kFusedBlock = RT_FUSED_BLOCK;  // fused Whitted kernels: 168 VGPRs per lane (428 B of scratch instead of 604 B) beat a fourth wave per SIMD
constexpr int kMaxLdsImage = 150 * 1024;
constexpr int kNF = 26;              // doubles per frame
constexpr int kMaxDepthLimit = 64;
constexpr int kCounterSlots = 32768;  // one row of 8 counters per wave: no same-address atomics
constexpr int kCounterRow = 8;
#ifndef RT_TRACE_WAVES_PER_EU
#define RT_TRACE_WAVES_PER_EU 4
#endif
#ifndef RT_WAVES_PER_EU
#define RT_WAVES_PER_EU 4  // 128 VGPRs: the walk's live set is small; shading spills a little instead
#endif

// primitive reference = type << 28 | index within the type's array; -1 = none
__host__ __device__ inline int make_ref(int type, int idx) { return (type << 28) | idx; }
RT_DEV int ref_type(int ref) { return ref >> 28; }
RT_DEV int ref_index(int ref) { return ref & 0x0FFFFFFF; }

struct DMaterial {
    double ks, kd, reflectivity, ior;
    double diffuse[3];
    double specular[3];
    int spec_exp, reflective, refractive, texture;
    double spec_brdf[3];  // specular * ks * ((exp + 2) / (2 pi)), the sample-independent part of RayTracer.cpp:624 / :668
};
struct DLight {
    int type, pad;
    double intensity[3], position[3], normal[3], u[3], v[3];
    double width, height;
};
struct DTexture {
    int width, height;
    long long offset;  // into the texel byte array
};

struct SceneView {
    int n_nodes, n_prims;
    int off_nodes, off_sph, off_tri, off_cyl;         // byte offsets into the image
    int off_gid_sph, off_gid_tri, off_gid_cyl;        // int tables: typed index -> primitive ID
    int off_ref;                                      // int table: primitive ID -> ref (linear scan)
    int off_matidx, off_mat, mat_staged;              // int table: primitive ID -> material; DMaterial records; records inside lds_bytes
    int image_bytes;
    int lds_bytes;  // prefix of the image the frame kernels stage into LDS (everything the chosen traversal touches per ray)
    int top_bytes;  // when that prefix does not fit: leading filter-BVH nodes (most visited first) staged instead, see StagedImg
    int n_lights;
    int regular;  // every coordinate is 0 or within [2^-300, 2^300]: shared-reciprocal division is exact
    int fbvh_ok, n_fnodes;                       // filter BVH present (fbvh_build.cpp)
    int off_fnodes, off_leafprim;                // 96 B nodes (box, leaf range, [octant] int2 links), 64 B leaf records
    int off_leafbox;                             // 32 B per leaf record: the primitive's own enlarged fp32 box, laid out like a node's (global copy only; pt_classify)
    double scene_scale;
    const unsigned char* image;  // global copy of the image
    const DMaterial* materials;
    const int* mat_of_gid;
    const DLight* lights;
    const DTexture* textures;
    const unsigned char* texels;
    double bg[3];
};

struct RenderArgs {
    // camera (src/Camera.cpp)
    double cam_pos[3], cam_ll[3], cam_hor[3], cam_ver[3], cam_u[3], cam_v[3];
    double lens_radius;
    double exposure, bias;
    int W, H;
    int mode, tonemap, max_depth;
    int pixel_samples, light_samples, sqrt_spp;
    int s_begin, s_end;
    int band_rows, band_rank, band_world;
    int local_rows;
    int tiles_x, total_items;
    unsigned int seed_lo, seed_hi;
    int accumulate_only;  // path trace: write raw sums
    int no_jitter;        // rt_params.debug_flags & RT_DEBUG_NO_JITTER: strata sampled at their lower-left corner
    // outputs / workspace
    double* rgb;              // fp64 frame (may be null when rgb8 is given: Whitted / binary only)
    unsigned char* rgb8;      // or null: the PPM payload, int(255.999 * c) per component (RayTracer.cpp:359-362)
    int* hit_id;
    double* frames;           // [max_depth+1][kNF][n_threads]
    long long frame_stride;   // n_threads
    unsigned int* ticket;
    unsigned long long* counters;  // rays, box, sphere, triangle, cylinder tests
};

// ------------------------------------------------------------------------------------------------
// Closest hit (Scene::intersect)
// ------------------------------------------------------------------------------------------------
struct HitT {
    double t;
    int ref;   // primitive ref or -1
    int part;  // cylinder part
};

struct Tally {
    unsigned rays, box, sph, tri, cyl;
    unsigned walks;  // traversals actually started (trace_ray): below `rays` where a ray's answer cannot reach the image and is not computed
    // Counting traversals (walk / cull) only: of the tests above, those of rays the PRODUCTION traversal would not walk --
    // shadow rays of a Whitted hit whose local colour is dead, of a path-traced hit that is dark or clear -- priced as
    // SURVEY.md 8(d) prices them (24 per box, 33 / 62 / 150 per sphere / triangle / cylinder test).
    unsigned unwalked_ops;
};
// a traversal that was counted into `t` on behalf of such a ray
RT_DEV void tally_unwalked(Tally& tally, const Tally& t) {
    tally.box += t.box;
    tally.sph += t.sph;
    tally.tri += t.tri;
    tally.cyl += t.cyl;
    tally.walks += t.walks;
    tally.unwalked_ops += 24u * t.box + 33u * t.sph + 62u * t.tri + 150u * t.cyl;
}

// Adds a wave's tallies to the row this wave owns (rows are per (block, wave); kernels on one
// stream run one after another, so a plain read-modify-write is enough).  4096 waves doing 5
// atomics each on the same 5 words cost ~250 us per launch; this costs nothing.
// `overwrite`: the first kernel of a frame stores instead of adding, which spares the host a 2 MB memset per
// frame (a fill kernel that has to wait for a free CU when several frames are in flight).
RT_DEV void flush_tally(const Tally& tally, unsigned long long* counters, bool overwrite = false) {
    unsigned long long c[7] = {tally.rays, tally.box, tally.sph, tally.tri, tally.cyl, tally.walks, tally.unwalked_ops};
    for (int off = 32; off > 0; off >>= 1)
        for (int k = 0; k < 7; ++k) c[k] += __shfl_down(c[k], off);
    if ((threadIdx.x & 63) == 0) {
        unsigned row = blockIdx.x * (blockDim.x >> 6) + (threadIdx.x >> 6);
        if (row < (unsigned)kCounterSlots) {
            unsigned long long* r = counters + (size_t)row * kCounterRow;
            if (overwrite)
                for (int k = 0; k < 7; ++k) r[k] = c[k];
            else
                for (int k = 0; k < 7; ++k) r[k] += c[k];
        } else {
            for (int k = 0; k < 7; ++k)
                if (c[k]) atomicAdd(&counters[k], c[k]);
        }
    }
}

template <typename P>
RT_DEV bool prim_t(P img, const SceneView& sv, int ref, D3 o, D3 d, double& t, int& part, Tally& tally) {
    int type = ref_type(ref), idx = ref_index(ref);
    part = 0;
    if (type == RT_SPHERE) ++tally.sph;
    else if (type == RT_TRIANGLE) ++tally.tri;
    else ++tally.cyl;
    if (type == RT_SPHERE) return sphere_t((const double*)(img + sv.off_sph) + idx * kSphereDoubles, o, d, t);
    if (type == RT_TRIANGLE) return triangle_t((const double*)(img + sv.off_tri) + idx * kTriDoubles, o, d, t);
    return cylinder_t((const double*)(img + sv.off_cyl) + idx * kCylDoubles, o, d, t, part);
}

template <typename P>
RT_DEV HitT closest_hit_ref(P img, const SceneView& sv, D3 o, D3 d, Tally& tally) {
    HitT best;
    best.ref = -1;
    best.part = 0;
    if (sv.n_nodes > 0) {
        // BVHNode::intersect as a fixed pre-order walk; "right wins unless left.t < right.t"
        // == the last leaf among the minimal t.
        best.t = __builtin_inf();
        int i = 0;
        while (i < sv.n_nodes) {
            const double* nb = (const double*)(img + sv.off_nodes + i * kNodeBytes);
            const int* ni = (const int*)(nb + 6);
            int skip = ni[0], r0 = ni[1], r1 = ni[2];
            ++tally.box;
            bool hit = box_hit(nb, o, d);
            if (hit && r0 >= 0) {
                double t;
                int part;
                if (prim_t(img, sv, r0, o, d, t, part, tally) && !(best.t < t)) {
                    best.t = t;
                    best.ref = r0;
                    best.part = part;
                }
                if (r1 >= 0) {
                    if (prim_t(img, sv, r1, o, d, t, part, tally) && !(best.t < t)) {
                        best.t = t;
                        best.ref = r1;
                        best.part = part;
                    }
                }
            }
            i = hit ? i + 1 : skip;
        }
    } else {
        // linear scan (Scene.cpp:67-79): strict <, first object wins
        best.t = 1.7976931348623157e308;  // numeric_limits<double>::max()
        const int* refs = (const int*)(img + sv.off_ref);
        for (int k = 0; k < sv.n_prims; ++k) {
            int r = refs[k];
            double t;
            int part;
            if (prim_t(img, sv, r, o, d, t, part, tally) && t < best.t) {
                best.t = t;
                best.ref = r;
                best.part = part;
            }
        }
    }
    return best;
}

// The same walk for "regular" rays (device_math.hpp: ray_regular), restructured for the SIMD:
//   * quotients by the three direction components share one refined reciprocal each (3
//     instructions per exact division instead of 11), and the slab logic is min/max;
//   * while-while form: every lane walks until IT holds a leaf, then all lanes run the primitive
//     tests together, instead of every walk step paying for whichever primitive types some lane
//     happened to reach.
// Visit order per lane, tie rule and every computed t are those of closest_hit_ref (the parity tests
// compare the two bit for bit); irregular rays (a zero direction component, ...) take the literal
// walk.
// kCull: skip nodes that provably cannot hold the winner (entry distance beyond the best t so far,
//        with the node's safety margin; nodes whose primitives may stick out of their boxes -- the
//        reference's cylinder boxes assume a +Y axis -- carry an infinite margin and are never culled).
// kAny:  shadow query "is there a hit with t < t_limit": stops at the first one.  Equivalent to the
//        reference's "closest hit, then t < lightDistance" (RayTracer.cpp:725-729) because the
//        closest t is below the limit iff some t is.  Returns ref >= 0 iff occluded.
template <bool kCull, bool kAny, typename P>
RT_DEV HitT closest_hit_fast(P img, const SceneView& sv, D3 o, D3 d, Tally& tally, double t_limit = 0.0) {
    if (!(sv.regular && ray_regular(o, d))) {
        HitT h = closest_hit_ref(img, sv, o, d, tally);
        if (kAny && !(h.ref >= 0 && h.t < t_limit)) h.ref = -1;
        return h;
    }
    HitT best;
    best.ref = -1;
    best.part = 0;
    const double a = dot(d, d);  // Sphere.cpp:16, the same value for every sphere of this ray
    const double two_a = 2.0 * a;
    const double r2a = refine_rcp(two_a);
    const double* sph = (const double*)(img + sv.off_sph);
    const double* tri = (const double*)(img + sv.off_tri);
    const double* cyl = (const double*)(img + sv.off_cyl);
    if (sv.n_nodes > 0) {
        const D3 r = mk(refine_rcp(d.x), refine_rcp(d.y), refine_rcp(d.z));
        best.t = __builtin_inf();
        int i = 0;
        // node count in a VGPR: as an SGPR the compiler re-loads it from the kernarg segment inside
        // the walk (s_load + wait on every missed box)
        int n_nodes;
        asm volatile("v_mov_b32 %0, %1" : "=v"(n_nodes) : "s"(sv.n_nodes));
        for (;;) {
            int p0 = -1, p1 = -1;
            while (i < n_nodes) {
                const double* nb = (const double*)(img + sv.off_nodes + i * kNodeBytes);
                const int* ni = (const int*)(nb + 6);
                int skip = ni[0], r0 = ni[1], r1 = ni[2];
                ++tally.box;
                double t_enter;
                bool hit = box_hit_regular(nb, o, d, r, t_enter);
                if (kCull) {
                    // nothing inside this box can beat `bound`: hits in it have t >= t_enter up to the
                    // rounding of the primitive tests, which the margin covers many times over
                    double bound = kAny ? t_limit : best.t;
                    double margin = (double)__int_as_float(ni[3]) + 1e-4 * fabs(t_enter);
                    if (t_enter - margin > bound) hit = false;
                }
                i = hit ? i + 1 : skip;
                if (hit && r0 >= 0) {
                    p0 = r0;
                    p1 = r1;
                    break;
                }
            }
            if (p0 < 0) break;
#pragma unroll 1
            for (int k = 0; k < 2; ++k) {
                int ref = k ? p1 : p0;
                if (ref < 0) break;
                int type = ref_type(ref), idx = ref_index(ref);
                double t;
                int part = 0;
                bool ok;
                if (type == RT_SPHERE) {
                    ++tally.sph;
                    ok = sphere_t_shared(sph + idx * kSphereDoubles, o, d, a, two_a, r2a, t);
                } else if (type == RT_TRIANGLE) {
                    ++tally.tri;
                    ok = triangle_t(tri + idx * kTriDoubles, o, d, t);
                } else {
                    ++tally.cyl;
                    ok = cylinder_t(cyl + idx * kCylDoubles, o, d, t, part);
                }
                if (kAny) {
                    if (ok && t < t_limit) {
                        best.t = t;
                        best.ref = ref;
                        i = n_nodes;  // occluded: leave the walk
                        break;
                    }
                } else if (ok && !(best.t < t)) {
                    best.t = t;
                    best.ref = ref;
                    best.part = part;
                }
            }
        }
    } else {
        best.t = 1.7976931348623157e308;
        const int* refs = (const int*)(img + sv.off_ref);
        for (int k = 0; k < sv.n_prims; ++k) {
            int ref = refs[k];
            int type = ref_type(ref), idx = ref_index(ref);
            double t;
            int part = 0;
            bool ok;
            if (type == RT_SPHERE) {
                ++tally.sph;
                ok = sphere_t_shared(sph + idx * kSphereDoubles, o, d, a, two_a, r2a, t);
            } else if (type == RT_TRIANGLE) {
                ++tally.tri;
                ok = triangle_t(tri + idx * kTriDoubles, o, d, t);
            } else {
                ++tally.cyl;
                ok = cylinder_t(cyl + idx * kCylDoubles, o, d, t, part);
            }
            if (kAny) {
                if (ok && t < t_limit) {
                    best.t = t;
                    best.ref = ref;
                    break;
                }
            } else if (ok && t < best.t) {
                best.t = t;
                best.ref = ref;
                best.part = part;
            }
        }
    }
    return best;
}

// Scene image of a scene too large for LDS: everything is read from the global copy (L1 / L2), except that the
// first top_bytes of the image -- the leading, most visited filter-BVH nodes (fbvh_build.cpp orders them by
// surface area) -- also sit in LDS.  Behaves like the plain pointer for `img + offset`; only the node fetch of
// closest_hit_fbvh asks where a node lives.
struct StagedImg {
    const unsigned char* base;  // global copy of the whole image
    const unsigned char* top;   // LDS copy of its first top_bytes
    int top_bytes;
    RT_DEV const unsigned char* operator+(long long off) const { return base + off; }
};
RT_DEV const unsigned char* fnode_ptr(const unsigned char* img, int off) { return img + off; }
RT_DEV const unsigned char* fnode_ptr(const StagedImg& img, int off) { return (off < img.top_bytes ? img.top : img.base) + off; }
RT_DEV const unsigned char* global_image(const unsigned char* img) { return img; }
RT_DEV const unsigned char* global_image(const StagedImg& img) { return img.base; }

// ---- filter-BVH traversal (fbvh_build.cpp explains why this returns the reference's answer) ------
// Inner loop: fp32 conservative slab test on an enlarged box, children in near-to-far order for the
// ray's octant through precomputed links (no stack), culled against the best t so far.  Leaves: the
// reference's own two fp64 tests -- the gate box (BoundingBox::intersect on the BVHNode that holds
// the primitive) and the primitive -- and the reference's tie rule via the DFS-order key.
// Box / primitive test counters of the filter-BVH traversal: off in the production build (one VALU
// instruction per walk step and four live VGPRs; rt_stats then reports rays only for this traversal --
// the walk / cull traversals always count).  -DRT_FBVH_COUNTS=1 turns them on.
#ifndef RT_FBVH_COUNTS
#define RT_FBVH_COUNTS 0
#endif
template <bool kAny, typename P>
RT_DEV HitT closest_hit_fbvh(P img, const SceneView& sv, D3 o, D3 d, Tally& tally, double t_limit = 0.0) {
    // Which rays take the filter BVH is a routing decision only (both walks return the same hit): direction
    // components in [2^-60, 2), origin components 0 or at least 2^-300 and within 4 scene sizes (<= 2^300) -- inside
    // ray_regular()'s windows, tested on the exponent fields: 19 vector instructions where the double compares
    // took 32 and five branches, paid by every ray.
    const double reach = 4.0 * sv.scene_scale;
    const unsigned edx = (unsigned)__double2hiint(d.x) & 0x7FFFFFFFu, edy = (unsigned)__double2hiint(d.y) & 0x7FFFFFFFu,
                   edz = (unsigned)__double2hiint(d.z) & 0x7FFFFFFFu;
    const unsigned eox = (unsigned)__double2hiint(o.x) & 0x7FFFFFFFu, eoy = (unsigned)__double2hiint(o.y) & 0x7FFFFFFFu,
                   eoz = (unsigned)__double2hiint(o.z) & 0x7FFFFFFFu;
    const bool dir_ok = min(min(edx, edy), edz) >= 0x3C300000u && max(max(edx, edy), edz) < 0x40000000u;
    const bool org_ok = ((eox >= 0x2D300000u) | (o.x == 0.0)) & ((eoy >= 0x2D300000u) | (o.y == 0.0)) &
                        ((eoz >= 0x2D300000u) | (o.z == 0.0)) & (fabs(o.x) <= reach) & (fabs(o.y) <= reach) &
                        (fabs(o.z) <= reach);
    const bool eligible = sv.fbvh_ok && sv.regular && reach <= 0x1p300 && dir_ok && org_ok;
    if (!eligible) {  // reference nodes: global copy only
        if (RT_FBVH_COUNTS) return closest_hit_fast<false, kAny>(sv.image, sv, o, d, tally, t_limit);
        Tally unused = {0, 0, 0, 0, 0};
        return closest_hit_fast<false, kAny>(sv.image, sv, o, d, unused, t_limit);
    }
    const double a = dot(d, d);
    const double two_a = 2.0 * a;
    const double r2a = refine_rcp(two_a);
    const D3 r = mk(refine_rcp(d.x), refine_rcp(d.y), refine_rcp(d.z));
    // fp32 slab test as one FMA per plane: t = b * (1/d) - o * (1/d).  Error in position space is
    // (|b| + |o|) 2^-24 <= 5 S 2^-24 = 3e-7 S, inside the 2^-20 S the boxes were enlarged by.
    const float rix = (float)r.x, riy = (float)r.y, riz = (float)r.z;
    const float nox = -(float)(o.x * r.x), noy = -(float)(o.y * r.y), noz = -(float)(o.z * r.z);
    const int oct = (d.x < 0 ? 1 : 0) | (d.y < 0 ? 2 : 0) | (d.z < 0 ? 4 : 0);
    const int link_off = 32 + oct * 8;
    const double* sph = (const double*)(img + sv.off_sph);
    const double* tri = (const double*)(img + sv.off_tri);
    const double* cyl = (const double*)(img + sv.off_cyl);
    HitT best;
    best.ref = -1;
    best.part = 0;
    best.t = kAny ? t_limit : __builtin_inf();
    int best_key = -2147483647 - 1;
    float bound = kAny ? (float)(t_limit * 1.00003) : __builtin_inff();
    // enter (1 - s) <= exit (1 + s) with s = 2e-5 [fp32 rounding (1e-6) + hit points displaced along the
    // ray (1e-5 t)], written as max(enter, 0) * (1 - 2.1 s) <= exit: the same for enter > 0, and for
    // enter <= 0 both reduce to exit >= 0.
    const float kShrink = 1.0f - 4.2e-5f;
    // `cur`: byte offset of a node record in the image; -1: done; < -1: a leaf to process,
    // -2 - (first record << 2 | records), which is what a leaf's "pass" links hold -- so the walk loop has
    // one exit test and never reads a leaf count.
    int cur = sv.off_fnodes;
    for (;;) {
        int after = -1;  // where the walk goes on after a leaf: the leaf's "fail" link
        while (cur >= 0) {
            const unsigned char* np = fnode_ptr(img, cur);
            const float4 b0 = *(const float4*)np;              // lo.x hi.x lo.y hi.y
            const float2 b1 = *(const float2*)(np + 16);       // lo.z hi.z
            const int2 lk = *(const int2*)(np + link_off);
            if (RT_FBVH_COUNTS) ++tally.box;
            const float x0 = __builtin_fmaf(b0.x, rix, nox), x1 = __builtin_fmaf(b0.y, rix, nox);
            const float y0 = __builtin_fmaf(b0.z, riy, noy), y1 = __builtin_fmaf(b0.w, riy, noy);
            const float z0 = __builtin_fmaf(b1.x, riz, noz), z1 = __builtin_fmaf(b1.y, riz, noz);
            float enter = fmaxf(fmaxf(fminf(x0, x1), fminf(y0, y1)), fminf(z0, z1));
            float exit = fminf(fminf(fmaxf(x0, x1), fmaxf(y0, y1)), fmaxf(z0, z1));
            bool fail = fmaxf(enter, 0.f) * kShrink > fminf(exit, bound);  // a NaN passes
            after = lk.y;
            cur = fail ? lk.y : lk.x;
        }
        if (cur == -1) break;
        const int code = -2 - cur;
        const int lf = code >> 2, lc = code & 3;
        cur = after;
#pragma unroll 1
        for (int k = 0; k < lc; ++k) {
            const unsigned char* rec = img + sv.off_leafprim + (lf + k) * kLeafBytes;
            int4 lp = *(const int4*)rec;  // ref, has gate, key; then the gate box (6 doubles)
            if (lp.y >= 0) {
                const double* nb = (const double*)(rec + 16);
                double t_enter;
                if (RT_FBVH_COUNTS) ++tally.box;
                if (!box_hit_regular(nb, o, d, r, t_enter)) continue;
            }
            int ref = lp.x;
            int type = ref_type(ref), idx = ref_index(ref);
            double t;
            int part = 0;
            bool ok;
            if (type == RT_SPHERE) {
                if (RT_FBVH_COUNTS) ++tally.sph;
                ok = sphere_t_shared(sph + idx * kSphereDoubles, o, d, a, two_a, r2a, t);
            } else if (type == RT_TRIANGLE) {
                if (RT_FBVH_COUNTS) ++tally.tri;
                ok = triangle_t(tri + idx * kTriDoubles, o, d, t);
            } else {
                if (RT_FBVH_COUNTS) ++tally.cyl;
                ok = cylinder_t(cyl + idx * kCylDoubles, o, d, t, part);
            }
            if (!ok) continue;
            if (kAny) {
                if (t < t_limit) {
                    best.t = t;
                    best.ref = ref;
                    cur = -1;  // occluded: leave the walk
                    break;
                }
            } else if (t < best.t || (t == best.t && lp.z > best_key)) {
                best.t = t;
                best.ref = ref;
                best.part = part;
                best_key = lp.z;
                bound = (float)(t * 1.00003);
            }
        }
    }
    if (kAny && best.ref < 0) best.t = 0.0;
    return best;
}

// kTrav: 1 = visit-for-visit walk of the reference tree, 2 = + culling / any-hit, 3 = filter BVH
template <int kTrav, bool kAny, typename P>
RT_DEV HitT trace_ray(P img, const SceneView& sv, D3 o, D3 d, Tally& tally, double t_limit = 0.0) {
    if (kTrav != 3 || RT_FBVH_COUNTS) ++tally.walks;  // (like the test counters: compiled out of the production traversal)
    if (kTrav == 3) return closest_hit_fbvh<kAny>(img, sv, o, d, tally, t_limit);
    if (kTrav == 2) return closest_hit_fast<true, kAny>(img, sv, o, d, tally, t_limit);
    HitT h = closest_hit_fast<false, false>(img, sv, o, d, tally);
    if (kAny && !(h.ref >= 0 && h.t < t_limit)) h.ref = -1;
    return h;
}

template <bool kFast, typename P>
RT_DEV HitT closest_hit(P img, const SceneView& sv, D3 o, D3 d, Tally& tally) {
    if (kFast) return closest_hit_fast<false, false>(img, sv, o, d, tally);
    return closest_hit_ref(img, sv, o, d, tally);
}

template <typename P>
RT_DEV int gid_of_ref(P img, const SceneView& sv, int ref) {
    int type = ref_type(ref), idx = ref_index(ref);
    int off = type == RT_SPHERE ? sv.off_gid_sph : (type == RT_TRIANGLE ? sv.off_gid_tri : sv.off_gid_cyl);
    return ((const int*)(img + off))[idx];
}

// Material of a primitive: two reads of the scene image (LDS when it is staged) instead of a dependent
// chain of global loads per shaded hit.
template <typename P>
RT_DEV int material_index(P img, const SceneView& sv, int ref) {
    return ((const int*)(img + sv.off_matidx))[gid_of_ref(img, sv, ref)];
}
template <typename P>
RT_DEV const DMaterial& material_at(P img, const SceneView& sv, int mi) {
    // the records themselves are in the staged part only when they fit next to the geometry
    const unsigned char* base = sv.mat_staged ? img + 0 : sv.image;
    return *(const DMaterial*)(base + sv.off_mat + mi * (int)sizeof(DMaterial));
}
template <typename P>
RT_DEV const DMaterial& material_of(P img, const SceneView& sv, int ref) {
    return material_at(img, sv, material_index(img, sv, ref));
}

// point and normal of the winning primitive, computed from t exactly as the intersect functions
// fill the HitRecord (Sphere.cpp:38-40, Triangle.cpp:68-70, Cylinder.cpp:44-47/72/87/95-97).
template <typename P>
RT_DEV void hit_geometry(P img, const SceneView& sv, const HitT& h, D3 o, D3 d, D3& point, D3& normal) {
    int type = ref_type(h.ref), idx = ref_index(h.ref);
    point = o + d * h.t;
    if (type == RT_SPHERE) {
        const double* s = (const double*)(img + sv.off_sph) + idx * kSphereDoubles;
        normal = normalize(point - ld3(s));
    } else if (type == RT_TRIANGLE) {
        const double* tr = (const double*)(img + sv.off_tri) + idx * kTriDoubles;
        normal = ld3(tr + 9);
    } else {
        const double* cy = (const double*)(img + sv.off_cyl) + idx * kCylDoubles;
        D3 base = ld3(cy), axis = ld3(cy + 3);
        if (h.part == 0) {
            double y = dot(o + d * h.t - base, axis);
            normal = normalize(point - base - axis * y);
        } else if (h.part == 1) {
            normal = -axis;
        } else {
            normal = axis;
        }
    }
}

// getUV of the three classes (Sphere.cpp:50-57, Triangle.cpp:15-39, Cylinder.cpp:109-129)
template <typename P>
RT_DEV void prim_uv(P img, const SceneView& sv, int ref, D3 point, double& u, double& v) {
    int type = ref_type(ref), idx = ref_index(ref);
    if (type == RT_SPHERE) {
        const double* s = (const double*)(img + sv.off_sph) + idx * kSphereDoubles;
        D3 p = normalize(point - ld3(s));
        double phi = atan2(p.z, p.x);
        double theta = acos(p.y);
        u = (phi + M_PI) / (2 * M_PI);
        v = theta / M_PI;
    } else if (type == RT_TRIANGLE) {
        const double* tr = (const double*)(img + sv.off_tri) + idx * kTriDoubles;
        D3 e1 = ld3(tr + 3), e2 = ld3(tr + 6);
        D3 pv = point - ld3(tr);
        double d00 = dot(e1, e1), d01 = dot(e1, e2), d11 = dot(e2, e2);
        double d20 = dot(pv, e1), d21 = dot(pv, e2);
        double denom = d00 * d11 - d01 * d01;
        double vc = (d11 * d20 - d01 * d21) / denom;
        double wc = (d00 * d21 - d01 * d20) / denom;
        double uc = 1.0 - vc - wc;
        u = (uc + vc) / 2.0;
        v = (vc + wc) / 2.0;
    } else {
        const double* cy = (const double*)(img + sv.off_cyl) + idx * kCylDoubles;
        D3 axis = ld3(cy + 3);
        D3 p = point - ld3(cy);
        double y = dot(p, axis);
        D3 pp = p - axis * y;
        double theta = atan2(pp.z, pp.x);
        if (theta < 0) theta += 2 * M_PI;
        u = theta / (2 * M_PI);
        v = y / cy[7];
        v = std_clamp(v, 0.0, 1.0);
    }
}

// Material::getTextureColor (src/Material.cpp:69-86)
RT_DEV D3 texture_color(const SceneView& sv, const DMaterial& m, double u, double v) {
    if (m.texture < 0) return ld3(m.diffuse);
    DTexture t = sv.textures[m.texture];
    if (t.width <= 0 || t.height <= 0) return ld3(m.diffuse);
    u = u - floor(u);
    v = v - floor(v);
    int x = (int)(u * t.width);
    int y = (int)((1.0 - v) * t.height);
    x = x < 0 ? 0 : (x > t.width - 1 ? t.width - 1 : x);
    y = y < 0 ? 0 : (y > t.height - 1 ? t.height - 1 : y);
    const unsigned char* px = sv.texels + t.offset + ((long long)y * t.width + x) * 3;
    return mk(px[0] / 255.0, px[1] / 255.0, px[2] / 255.0);
}

#ifndef RT_TEXTURE_CALL
#define RT_TEXTURE_CALL 0
#endif
// The textured case as a real call (k_pathwave.hip: pt_kernel): getUV's atan2 / acos bring ~40 polynomial coefficients
// that the compiler otherwise hoists out of the kernel's loops into vector registers and spills (one scratch reload per
// coefficient and use, each waited for: final_textured_lens.json 40.7 -> 38.8 ms per 16 strata at 1080p; 41 instead of
// 99 spilled dwords).  Not for the fused Whitted kernels: 36 instead of 81 spilled dwords there too, but clutter_500
// renders 4 % slower with a call in the kernel (28.3 against 29.5 Grays/s), so they keep the inlined form.
template <typename P>
#if RT_TEXTURE_CALL
__device__ __attribute__((noinline))
#else
RT_DEV
#endif
D3 textured_colour(P img, const SceneView& sv, const DMaterial& m, int ref, D3 point) {
    double u, v;
    prim_uv(img, sv, ref, point, u, v);
    return texture_color(sv, m, u, v);
}
template <typename P>
RT_DEV D3 surface_colour(P img, const SceneView& sv, const DMaterial& m, int ref, D3 point) {
    if (m.texture < 0) return ld3(m.diffuse);
    return textured_colour(img, sv, m, ref, point);
}

// toneMap (RayTracer.cpp:161-202)
RT_DEV D3 uncharted2_partial(D3 x) {
    const double A = 0.15f, B = 0.50f, C = 0.10f, D = 0.20f, E = 0.02f, F = 0.30f;
    // float arithmetic on the literals as written: C*B, D*E, D*F, E/F are float products
    const double CB = (double)(0.10f * 0.50f), DE = (double)(0.20f * 0.02f), DF = (double)(0.20f * 0.30f),
                 EF = (double)(0.02f / 0.30f);
    (void)C; (void)D; (void)E; (void)F;
    return ((x * (x * A + CB) + DE) / (x * (x * A + B) + DF)) - EF;
}
RT_DEV D3 tone_map(D3 c, int tm) {
    if (tm == RT_TONEMAP_REINHARD) return c / (c + mk(1, 1, 1));
    if (tm == RT_TONEMAP_WARD) {
        c = c / (c + mk(1, 1, 1));
        return c * (c * (c * 0.4 + 0.4) + 1.219) / (c * (c * 0.3 + 0.5) + 2.219);
    }
    if (tm == RT_TONEMAP_UNCHARTED2) {
        D3 curr = uncharted2_partial(c * (double)2.0f);
        D3 W = mk((double)11.2f, (double)11.2f, (double)11.2f);
        D3 white = mk(1.0, 1.0, 1.0) / uncharted2_partial(W);
        return curr * white;
    }
    return c;
}
RT_DEV double clamp01(double x) { return std_min(1.0, std_max(0.0, x)); }

// ------------------------------------------------------------------------------------------------
// Frame stack
// ------------------------------------------------------------------------------------------------
// Field map (doubles).  Whitted and path trace share the first nine.
enum {
    F_POINT = 0,   // hit point
    F_NORMAL = 3,  // HitRecord::normal (unflipped)
    F_DIR = 6,     // direction of the ray that found the hit (Ray::direction)
    F_A = 9,       // Whitted: local colour          | PT: albedo (after Russian roulette)
    F_B = 12,      // Whitted: texture/diffuse colour | PT: direct light
    F_C = 15,      // Whitted: refracted colour       | PT: area-light accumulator
    F_D = 18,      // Whitted: diffuse accumulator    | PT: reflected colour (glass)
    F_E = 21,      // Whitted: specular accumulator (3) | PT: F_E fresnel, F_E+1 cosTheta
    F_I0 = 24,     // two packed ints: primitive ref | RNG node code (PT) / cylinder part
    F_I1 = 25      // two packed ints: stage | light index << 8 ... , sample index
};

struct Frames {
    double* base;
    long long stride;
    long long tid;
    RT_DEV double& at(int depth, int f) const { return base[((long long)depth * kNF + f) * stride + tid]; }
    RT_DEV void put3(int depth, int f, D3 v) const {
        at(depth, f) = v.x;
        at(depth, f + 1) = v.y;
        at(depth, f + 2) = v.z;
    }
    RT_DEV D3 get3(int depth, int f) const { return mk(at(depth, f), at(depth, f + 1), at(depth, f + 2)); }
    RT_DEV void puti(int depth, int f, int lo, int hi) const {
        long long v = ((long long)hi << 32) | (unsigned int)lo;
        at(depth, f) = __longlong_as_double(v);
    }
    RT_DEV void geti(int depth, int f, int& lo, int& hi) const {
        long long v = __double_as_longlong(at(depth, f));
        lo = (int)(v & 0xFFFFFFFFll);
        hi = (int)(v >> 32);
    }
};

enum { RAY_NONE = 0, RAY_CLOSEST = 1, RAY_SHADOW = 2 };

struct Lane {
    D3 o, d;       // current ray (d normalized by the Ray constructor rule)
    int kind;      // RAY_*
    int depth;     // recursion depth of the frame the current ray belongs to
    int px, lrow;  // pixel column, local row
    int sample;    // path trace: current stratum index
    bool exhausted;
};

RT_DEV void set_ray(Lane& L, D3 origin, D3 dir, int kind) {  // Ray::Ray normalizes (Ray.cpp:5-6)
    L.o = origin;
    L.d = normalize(dir);
    L.kind = kind;
}

RT_DEV int global_row(const RenderArgs& A, int lrow) {
    return ((lrow / A.band_rows) * A.band_world + A.band_rank) * A.band_rows + lrow % A.band_rows;
}

// Camera::getRay (src/Camera.cpp:33-49)
RT_DEV void camera_ray(const RenderArgs& A, double s, double t, bool use_lens, const Rng* rng, D3& origin, D3& dir) {
    D3 offset = mk(0, 0, 0);
    D3 pos = ld3(A.cam_pos);
    if (use_lens && A.lens_radius > 0.0) {
        double x, y;
        unsigned k = 0;
        do {  // randomInUnitDisk (include/Camera.h:37-46)
            x = -1.0 + 2.0 * rng_u(*rng, 0, 2 + 2 * k);
            y = -1.0 + 2.0 * rng_u(*rng, 0, 2 + 2 * k + 1);
            ++k;
        } while (x * x + y * y >= 1.0);
        D3 rd = mk(x, y, 0) * A.lens_radius;
        offset = ld3(A.cam_u) * rd.x + ld3(A.cam_v) * rd.y;
    }
    D3 image_point = ld3(A.cam_ll) + ld3(A.cam_hor) * s + ld3(A.cam_ver) * t;
    D3 rdir = image_point - pos - offset;
    origin = pos + offset;
    dir = normalize(rdir);  // normalized here and again by the Ray constructor
}

// ------------------------------------------------------------------------------------------------
// Whitted continuation (traceRay / computeShadingPhong, RayTracer.cpp:373-393, 694-804)
// ------------------------------------------------------------------------------------------------
enum { ST_LIGHTS = 0, ST_REFLECT = 1, ST_REFR_T = 2, ST_REFR_R = 3 };

// std::pow with a small non-negative integer exponent (Material::specularExponent is an int, and
// fresnelReflectance raises to 5.0) by square-and-multiply: ~10 multiplications instead of the ~250
// instructions of the general fp64 pow, two of which every shaded hit pays.  Not correctly rounded:
// the relative error is below n/2 ulp (1e-14 for n = 128), against the 1e-4 of the parity contract and
// the 1e-12 the tests hold Whitted frames to; ocml's own pow already differs from glibc's by <= 2 ulp.
// pow(x, 0) == 1 for every x, as in C.
RT_DEV double pow_int(double x, int n) {
    if (n < 0 || n > 4096) return pow(x, (double)n);
    double r = 1.0;
    for (unsigned k = (unsigned)n; k; k >>= 1) {
        if (k & 1u) r = r * x;
        x = x * x;
    }
    return r;
}
// The same products in the same order when every active lane of the wave has the same exponent (a scene's materials
// mostly share one): the loop then runs on the scalar unit and only the multiplications that count are issued --
// x^20 is 4 squarings and 2 products instead of 5 rounds of two multiplications, two selects, a shift and a compare.
RT_DEV double pow_int_wave(double x, int n) {
    const int n0 = __builtin_amdgcn_readfirstlane(n);
    if (__ballot(n != n0) != 0ull || n0 < 0 || n0 > 4096) return pow_int(x, n);
    double r = 1.0;
    for (unsigned k = (unsigned)n0; k; k >>= 1) {
        if (k & 1u) r = r * x;
        if (k > 1u) x = x * x;
    }
    return r;
}

RT_DEV double schlick(double cosTheta, double ior) {  // fresnelReflectance, RayTracer.cpp:500-504
    double r0 = (1.0 - ior) / (1.0 + ior);
    r0 = r0 * r0;
    double y = 1.0 - cosTheta, y2 = y * y;
    return r0 + (1.0 - r0) * (y2 * y2 * y);  // pow(y, 5.0)
}

// Final store of a Whitted / binary pixel (RayTracer.cpp:231-235 clamp, :359-362 PPM conversion): the fp64
// frame, the PPM bytes, or both.  `c` is the tone-mapped, exposed colour.
RT_DEV unsigned char ppm_byte(double c) {  // the same expression as quantize_kernel
    int v = (int)(255.999 * c);
    return (unsigned char)(v < 0 ? 0 : (v > 255 ? 255 : v));
}
RT_DEV void store_pixel(const RenderArgs& A, long long p, D3 c) {
    const double r = clamp01(c.x), g = clamp01(c.y), b = clamp01(c.z);
    if (A.rgb) {
        A.rgb[p * 3] = r;
        A.rgb[p * 3 + 1] = g;
        A.rgb[p * 3 + 2] = b;
    }
    if (A.rgb8) {
        A.rgb8[p * 3] = ppm_byte(r);
        A.rgb8[p * 3 + 1] = ppm_byte(g);
        A.rgb8[p * 3 + 2] = ppm_byte(b);
    }
}

RT_DEV D3 light_position(const DLight& Lt) {  // only PointLight overrides getPosition (Light.cpp:9-11)
    return Lt.type == RT_LIGHT_POINT ? ld3(Lt.position) : mk(0, 0, 0);
}

template <typename P>
RT_DEV void finish_pixel_whitted(const RenderArgs& A, Lane& L, D3 color) {
    color = tone_map(color, A.tonemap);
    color = color * A.exposure;
    store_pixel(A, (long long)L.lrow * A.W + L.px, color);
    L.kind = RAY_NONE;
}

// Runs after the lane's ray has been traced.  Leaves either a new ray in L (kind != NONE) or the
// pixel written (kind == NONE).
template <typename P>
RT_DEV void advance_whitted(P img, const SceneView& sv, const RenderArgs& A, const Frames& F, Lane& L, const HitT& h) {
    enum { PC_NEXT_LIGHT, PC_AFTER_LIGHTS, PC_AFTER_REFLECT, PC_RETURN };
    int pc;
    int depth = L.depth;
    // live frame
    D3 point, normal, dir, local, tex, diffuse, specular;
    int ref = -1, li = 0;
    D3 V = mk(0, 0, 0);
    const D3 bg = ld3(sv.bg);

    if (L.kind == RAY_SHADOW) {
        // back from the shadow ray of light `li` (RayTracer.cpp:721-743)
        point = F.get3(depth, F_POINT);
        normal = F.get3(depth, F_NORMAL);
        dir = F.get3(depth, F_DIR);
        tex = F.get3(depth, F_B);
        diffuse = F.get3(depth, F_D);
        specular = F.get3(depth, F_E);
        int stage_li, unused;
        F.geti(depth, F_I0, ref, unused);
        F.geti(depth, F_I1, stage_li, unused);
        li = stage_li >> 8;
        const DLight& Lt = sv.lights[li];
        D3 lpos = light_position(Lt);
        bool in_shadow = false;
        if (h.ref >= 0) {
            double ldist = length(lpos - point);
            if (h.t < ldist) in_shadow = true;
        }
        if (!in_shadow) {
            const DMaterial& m = material_of(img, sv, ref);
            D3 view = -normalize(dir);
            D3 ldir = normalize(lpos - point);
            D3 half = normalize(ldir + view);
            double df = std_max(0.0, dot(normal, ldir));
            diffuse = diffuse + tex * m.kd * df * ld3(Lt.intensity);
            double sf = pow_int(std_max(0.0, dot(normal, half)), m.spec_exp);
            specular = specular + ld3(m.specular) * m.ks * sf * ld3(Lt.intensity);
        }
        ++li;
        pc = PC_NEXT_LIGHT;
    } else {
        // back from a closest-hit ray at `depth` (traceRay, RayTracer.cpp:379-392)
        if (depth == 0 && A.hit_id) A.hit_id[(long long)L.lrow * A.W + L.px] = h.ref >= 0 ? gid_of_ref(img, sv, h.ref) : -1;
        if (h.ref < 0) {
            V = (A.mode == RT_MODE_BINARY) ? mk(0, 0, 0) : bg;
            pc = PC_RETURN;
        } else if (A.mode == RT_MODE_BINARY) {
            V = mk(1.0, 0.0, 0.0);
            pc = PC_RETURN;
        } else {
            ref = h.ref;
            dir = L.d;
            hit_geometry(img, sv, h, L.o, L.d, point, normal);
            const DMaterial& m = material_of(img, sv, ref);
            tex = surface_colour(img, sv, m, ref, point);
            diffuse = mk(0, 0, 0);
            specular = mk(0, 0, 0);
            li = 0;
            pc = PC_NEXT_LIGHT;
        }
    }

    for (;;) {
        if (pc == PC_NEXT_LIGHT) {
            if (li < sv.n_lights) {
                const DLight& Lt = sv.lights[li];
                D3 ldir = normalize(light_position(Lt) - point);
                F.put3(depth, F_POINT, point);
                F.put3(depth, F_NORMAL, normal);
                F.put3(depth, F_DIR, dir);
                F.put3(depth, F_B, tex);
                F.put3(depth, F_D, diffuse);
                F.put3(depth, F_E, specular);
                F.puti(depth, F_I0, ref, 0);
                F.puti(depth, F_I1, ST_LIGHTS | (li << 8), 0);
                L.depth = depth;
                set_ray(L, point + normal * A.bias, ldir, RAY_SHADOW);
                return;
            }
            local = tex * 0.25 + diffuse + specular;  // ambient + diffuse + specular (:705, :746)
            pc = PC_AFTER_LIGHTS;
        } else if (pc == PC_AFTER_LIGHTS) {
            const DMaterial& m = material_of(img, sv, ref);
            pc = PC_AFTER_REFLECT;
            if (m.reflective) {  // :749-760
                D3 n = normal;
                if (dot(dir, n) > 0.0) n = -n;
                D3 rdir = dir - n * 2.0 * dot(dir, n);
                if (depth + 1 >= A.max_depth) {
                    local = local * (1 - m.reflectivity) + bg * m.reflectivity;
                } else {
                    F.put3(depth, F_POINT, point);
                    F.put3(depth, F_NORMAL, normal);
                    F.put3(depth, F_DIR, dir);
                    F.put3(depth, F_A, local);
                    F.puti(depth, F_I0, ref, 0);
                    F.puti(depth, F_I1, ST_REFLECT, 0);
                    L.depth = depth + 1;
                    set_ray(L, point + n * A.bias, rdir, RAY_CLOSEST);
                    return;
                }
            }
        } else if (pc == PC_AFTER_REFLECT) {
            const DMaterial& m = material_of(img, sv, ref);
            V = local;
            pc = PC_RETURN;
            if (m.refractive && depth < A.max_depth) {  // :764-800
                double n1 = 1.0, n2 = m.ior;
                D3 n = normal;
                if (dot(dir, n) > 0.0) {
                    n = -n;
                    double s = n1; n1 = n2; n2 = s;
                }
                double eta = n1 / n2;
                double cosI = -dot(n, dir);
                double sinT2 = eta * eta * (1.0 - cosI * cosI);
                if (sinT2 <= 1.0) {
                    double cosT = sqrt(1.0 - sinT2);
                    D3 tdir = normalize(dir * eta + n * (eta * cosI - cosT));
                    double R = schlick(cosI, n2);
                    if (depth + 1 >= A.max_depth) {
                        V = bg * R + bg * (1.0 - R);
                    } else {
                        F.put3(depth, F_POINT, point);
                        F.put3(depth, F_NORMAL, normal);
                        F.put3(depth, F_DIR, dir);
                        F.at(depth, F_E) = R;
                        F.puti(depth, F_I0, ref, 0);
                        F.puti(depth, F_I1, ST_REFR_T, 0);
                        L.depth = depth + 1;
                        set_ray(L, point - n * A.bias, tdir, RAY_CLOSEST);
                        return;
                    }
                }
            }
        } else {  // PC_RETURN: V is the value of traceRay at `depth`
            if (depth == 0) {
                finish_pixel_whitted<P>(A, L, V);
                return;
            }
            --depth;
            int stage, unused;
            F.geti(depth, F_I1, stage, unused);
            F.geti(depth, F_I0, ref, unused);
            point = F.get3(depth, F_POINT);
            normal = F.get3(depth, F_NORMAL);
            dir = F.get3(depth, F_DIR);
            const DMaterial& m = material_of(img, sv, ref);
            if (stage == ST_REFLECT) {
                local = F.get3(depth, F_A);
                local = local * (1 - m.reflectivity) + V * m.reflectivity;
                pc = PC_AFTER_REFLECT;
            } else if (stage == ST_REFR_T) {
                // refracted colour is in; now the reflected ray (:792-795)
                D3 n = normal;
                if (dot(dir, n) > 0.0) n = -n;
                D3 rdir = dir - n * 2.0 * dot(dir, n);
                F.put3(depth, F_C, V);
                F.puti(depth, F_I1, ST_REFR_R, 0);
                L.depth = depth + 1;
                set_ray(L, point + n * A.bias, rdir, RAY_CLOSEST);
                return;
            } else {  // ST_REFR_R
                double R = F.at(depth, F_E);
                D3 tc = F.get3(depth, F_C);
                V = V * R + tc * (1.0 - R);
                pc = PC_RETURN;
            }
        }
    }
}

// ------------------------------------------------------------------------------------------------
// Path-trace continuation (traceRayPath / estimateDirectLight, RayTracer.cpp:506-689)
// ------------------------------------------------------------------------------------------------
enum { PT_LIGHTS = 0, PT_MIRROR = 1, PT_GLASS_R = 2, PT_GLASS_T = 3, PT_DIFFUSE = 4 };
enum { DIM_RR = 0, DIM_HEMI_1 = 2, DIM_HEMI_2 = 3, DIM_LIGHT0 = 4 };  // (the two hemisphere dimensions share one counter block)

RT_DEV D3 reflect_v(D3 v, D3 n) { return v - n * 2.0 * dot(v, n); }  // RayTracer.cpp:477-481
RT_DEV D3 refract_v(D3 I, D3 N, double eta_t, double eta_i) {        // RayTracer.cpp:430-448
    double cosi = -std_max(-1.0, std_min(1.0, dot(I, N)));
    if (cosi < 0) {  // one level of the reference's recursion: -N, swapped indices
        N = -N;
        double s = eta_t; eta_t = eta_i; eta_i = s;
        cosi = -std_max(-1.0, std_min(1.0, dot(I, N)));
        // a second flip cannot happen for finite inputs (cosi would have to be < 0 both ways)
    }
    double eta = eta_i / eta_t;
    double k = 1 - eta * eta * (1 - cosi * cosi);
    if (k < 0) return mk(0, 0, 0);
    return I * eta + N * (eta * cosi - sqrt(k));
}
RT_DEV double fresnel_f(D3 I, D3 N, double eta_t, double eta_i) {  // RayTracer.cpp:453-472
    double cosi = std_clamp(dot(I, N), -1.0, 1.0);
    double etai = eta_i, etat = eta_t;
    if (cosi > 0) { double s = etai; etai = etat; etat = s; }
    double sint = etai / etat * sqrt(std_max(0.0, 1 - cosi * cosi));
    if (sint >= 1) return 1.0;
    double cost = sqrt(std_max(0.0, 1 - sint * sint));
    cosi = fabs(cosi);
    double Rs = ((etat * cosi) - (etai * cost)) / ((etat * cosi) + (etai * cost));
    double Rp = ((etai * cosi) - (etat * cost)) / ((etai * cosi) + (etat * cost));
    return (Rs * Rs + Rp * Rp) / 2.0;
}
RT_DEV D3 hemisphere(D3 n, double r1, double r2) {  // randomInHemisphere, RayTracer.cpp:398-424
    double sinTheta = sqrt(1 - r1 * r1);
    double phi = 2 * M_PI * r2;
    double x = cos(phi) * sinTheta;
    double y = r1;
    double z = sin(phi) * sinTheta;
    D3 tangent;
    if (fabs(n.x) > fabs(n.y)) tangent = normalize(mk(-n.z, 0, n.x));
    else tangent = normalize(mk(0, n.z, -n.y));
    D3 bitangent = cross(n, tangent);
    D3 dir = tangent * x + n * y + bitangent * z;
    return normalize(dir);
}

// AreaLight::sample (src/AreaLight.cpp:3-20) for sample `si` of light `li` at RNG node `node`
template <bool kShared = false>  // kShared: ldir / distance through div3 (same bits)
RT_DEV void area_sample(const RenderArgs& A, const DLight& Lt, const Rng& rng, unsigned node, int li, int si, D3 point,
                        D3& ldir, double& distance, double& pdf) {
    double u0, u1;  // dimensions DIM_LIGHT0 + 2 m and + 2 m + 1 of sample m = li * light_samples + si: one counter block
    rng_u2(rng, node, DIM_LIGHT0 / 2 + (unsigned)(li * A.light_samples + si), u0, u1);
    double su = (u0 - 0.5) * Lt.width;
    double sv = (u1 - 0.5) * Lt.height;
    D3 sp = ld3(Lt.position) + ld3(Lt.u) * su + ld3(Lt.v) * sv;
    ldir = sp - point;
    distance = length(ldir);
    ldir = kShared ? div3(ldir, distance) : ldir / distance;
    double area = Lt.width * Lt.height;
    double cosine = std_max(0.0, dot(ld3(Lt.normal), -ldir));
    pdf = (distance * distance) / (area * cosine);
}

template <typename P>
RT_DEV D3 brdf_eval(P img, const SceneView& sv, const DMaterial& m, int ref, D3 point, D3 normal, D3 ldir, D3 view) {
    D3 dbrdf = (ld3(m.diffuse) * m.kd) / M_PI;
    if (m.texture >= 0) {
        double u, v;
        prim_uv(img, sv, ref, point, u, v);
        dbrdf = texture_color(sv, m, u, v) * m.kd / M_PI;
    }
    D3 half = normalize(ldir + view);
    double ndoth = std_max(0.0, dot(normal, half));
    double sf = pow_int(ndoth, m.spec_exp);
    D3 sbrdf = ld3(m.specular) * m.ks * ((m.spec_exp + 2.0) / (2.0 * M_PI)) * sf;
    return dbrdf + sbrdf;
}

// Starts stratum L.sample of the lane's pixel (RayTracer.cpp:270-283) or finishes the pixel.
RT_DEV void pt_next_sample(const RenderArgs& A, const Frames& F, Lane& L, Rng& rng) {
    const int pix_slot = A.max_depth;  // frame level used for the per-pixel colour sum
    if (L.sample >= A.s_end) {
        D3 color = F.get3(pix_slot, 0);
        long long idx = ((long long)L.lrow * A.W + L.px) * 3;
        if (A.accumulate_only) {
            A.rgb[idx] = color.x;
            A.rgb[idx + 1] = color.y;
            A.rgb[idx + 2] = color.z;
        } else {  // RayTracer.cpp:301-315
            color = color / (double)A.pixel_samples;
            color = tone_map(color, A.tonemap);
            color = color * A.exposure;
            color.x = pow(color.x, 1.0 / 2.2);
            color.y = pow(color.y, 1.0 / 2.2);
            color.z = pow(color.z, 1.0 / 2.2);
            A.rgb[idx] = clamp01(color.x);
            A.rgb[idx + 1] = clamp01(color.y);
            A.rgb[idx + 2] = clamp01(color.z);
        }
        L.kind = RAY_NONE;
        return;
    }
    if (A.max_depth <= 0) {  // traceRayPath(ray, 0) returns 0 without tracing (RayTracer.cpp:507-509)
        L.sample = A.s_end;
        pt_next_sample(A, F, L, rng);
        return;
    }
    int s = L.sample;
    int sx = s % A.sqrt_spp, sy = s / A.sqrt_spp;
    rng.sample = (unsigned)s;
    int j = global_row(A, L.lrow);
    double r1 = (sx + (A.no_jitter ? 0.0 : rng_u(rng, 0, 0))) / A.sqrt_spp;
    double r2 = (sy + (A.no_jitter ? 0.0 : rng_u(rng, 0, 1))) / A.sqrt_spp;
    double u = 1.0 - ((double)L.px + r1) / (A.W - 1);
    double v = ((double)j + r2) / (A.H - 1);
    D3 origin, dir;
    camera_ray(A, u, v, true, &rng, origin, dir);
    L.depth = 0;
    set_ray(L, origin, dir, RAY_CLOSEST);
}

template <typename P>
RT_DEV void advance_path(P img, const SceneView& sv, const RenderArgs& A, const Frames& F, Lane& L, Rng& rng,
                         const HitT& h) {
    enum { PC_LIGHT_LOOP, PC_MATERIAL, PC_RETURN };
    int pc;
    int depth = L.depth;
    D3 point, normal, dir, albedo, direct, acc;
    int ref = -1, li = 0, si = 0;
    unsigned node = 1;
    D3 V = mk(0, 0, 0);
    const D3 bg = ld3(sv.bg);

    if (L.kind == RAY_SHADOW) {
        point = F.get3(depth, F_POINT);
        normal = F.get3(depth, F_NORMAL);
        dir = F.get3(depth, F_DIR);
        albedo = F.get3(depth, F_A);
        direct = F.get3(depth, F_B);
        acc = F.get3(depth, F_C);
        int inode, packed;
        F.geti(depth, F_I0, ref, inode);
        node = (unsigned)inode;
        F.geti(depth, F_I1, packed, si);
        li = packed >> 8;
        const DLight& Lt = sv.lights[li];
        const DMaterial& m = material_of(img, sv, ref);
        D3 view = -normalize(dir);
        if (Lt.type == RT_LIGHT_POINT) {  // RayTracer.cpp:602-637
            D3 lpos = ld3(Lt.position);
            D3 ldir = normalize(lpos - point);
            double distance = length(lpos - point);
            bool shadowed = (h.ref >= 0) && (h.t < distance);
            if (!shadowed) {
                double ndotl = std_max(0.0, dot(normal, ldir));
                D3 brdf = brdf_eval(img, sv, m, ref, point, normal, ldir, view);
                direct = direct + brdf * ld3(Lt.intensity) * ndotl;
            }
            ++li;
            si = 0;
        } else {  // RayTracer.cpp:643-684
            D3 ldir;
            double distance, pdf;
            area_sample(A, Lt, rng, node, li, si, point, ldir, distance, pdf);
            bool shadowed = (h.ref >= 0) && (h.t < distance);
            if (!shadowed) {
                double ndotl = std_max(0.0, dot(normal, ldir));
                double ndotl_light = std_max(0.0, dot(ld3(Lt.normal), -ldir));
                if (ndotl > 0 && ndotl_light > 0) {
                    D3 brdf = brdf_eval(img, sv, m, ref, point, normal, ldir, view);
                    acc = acc + brdf * ld3(Lt.intensity) * ndotl * ndotl_light / pdf;
                }
            }
            ++si;
            if (si >= A.light_samples) {
                direct = direct + acc / (double)A.light_samples;
                acc = mk(0, 0, 0);
                si = 0;
                ++li;
            }
        }
        pc = PC_LIGHT_LOOP;
    } else {
        // back from a closest-hit ray: traceRayPath body up to the direct light (:511-539)
        if (depth == 0) node = 1;
        else {
            // the node code of this hit was stored by the parent when it launched the ray
            int lo, hi;
            F.geti(depth, F_I0, lo, hi);
            node = (unsigned)hi;
        }
        if (h.ref < 0) {
            V = bg;
            pc = PC_RETURN;
        } else {
            ref = h.ref;
            dir = L.d;
            hit_geometry(img, sv, h, L.o, L.d, point, normal);
            const DMaterial& m = material_of(img, sv, ref);
            albedo = surface_colour(img, sv, m, ref, point);
            pc = PC_LIGHT_LOOP;
            if (depth > 3) {  // Russian roulette (:530-536)
                double maxR = std_max(albedo.x, std_max(albedo.y, albedo.z));
                if (rng_u(rng, node, DIM_RR) > maxR) {
                    V = mk(0, 0, 0);
                    pc = PC_RETURN;
                } else {
                    albedo = albedo / maxR;
                }
            }
            direct = mk(0, 0, 0);
            acc = mk(0, 0, 0);
            li = 0;
            si = 0;
        }
    }

    for (;;) {
        if (pc == PC_LIGHT_LOOP) {
            // skip area lights with no samples (the reference's loop body never runs; 0/0 follows)
            if (li < sv.n_lights && sv.lights[li].type == RT_LIGHT_AREA && A.light_samples <= 0) {
                direct = direct + acc / (double)A.light_samples;
                ++li;
                continue;
            }
            if (li < sv.n_lights) {
                const DLight& Lt = sv.lights[li];
                D3 ldir;
                if (Lt.type == RT_LIGHT_POINT) {
                    ldir = normalize(ld3(Lt.position) - point);
                } else {
                    double distance, pdf;
                    area_sample(A, Lt, rng, node, li, si, point, ldir, distance, pdf);
                }
                F.put3(depth, F_POINT, point);
                F.put3(depth, F_NORMAL, normal);
                F.put3(depth, F_DIR, dir);
                F.put3(depth, F_A, albedo);
                F.put3(depth, F_B, direct);
                F.put3(depth, F_C, acc);
                F.puti(depth, F_I0, ref, (int)node);
                F.puti(depth, F_I1, PT_LIGHTS | (li << 8), si);
                L.depth = depth;
                set_ray(L, point + normal * A.bias, ldir, RAY_SHADOW);
                return;
            }
            pc = PC_MATERIAL;
        } else if (pc == PC_MATERIAL) {
            const DMaterial& m = material_of(img, sv, ref);
            D3 nf = normal;
            if (dot(dir, nf) > 0) nf = -nf;
            bool child_possible = depth + 1 < A.max_depth;  // else traceRayPath returns 0 with no ray
            // common part of the frame
            F.put3(depth, F_POINT, point);
            F.put3(depth, F_NORMAL, normal);
            F.put3(depth, F_DIR, dir);
            F.put3(depth, F_A, albedo);
            F.put3(depth, F_B, direct);
            if (m.reflective) {  // :543-549
                D3 rdir = normalize(reflect_v(normalize(dir), nf));
                if (!child_possible) {
                    V = direct + mk(0, 0, 0) * m.reflectivity;
                    pc = PC_RETURN;
                    continue;
                }
                F.puti(depth, F_I0, ref, (int)node);
                F.puti(depth, F_I1, PT_MIRROR, 0);
                F.puti(depth + 1, F_I0, 0, (int)(2 * node));
                L.depth = depth + 1;
                set_ray(L, point + nf * A.bias, rdir, RAY_CLOSEST);
                return;
            } else if (m.refractive) {  // :551-584
                D3 n = normal;
                double eta_i = 1.0, eta_t = m.ior;
                D3 incident = normalize(dir);
                bool entering = dot(incident, n) < 0;
                if (!entering) {
                    double s = eta_i; eta_i = eta_t; eta_t = s;
                    n = -n;
                }
                D3 bias = n * A.bias;
                D3 rdir = normalize(reflect_v(incident, n));
                if (!child_possible) {
                    // both children return 0 without tracing
                    double Fr = fresnel_f(incident, n, eta_t, eta_i);
                    D3 tdir = normalize(refract_v(incident, n, eta_t, eta_i));
                    D3 zero = mk(0, 0, 0);
                    D3 indirect = (length(tdir) > 0.0) ? (zero * Fr + zero * (1.0 - Fr)) : zero;
                    V = direct + indirect;
                    pc = PC_RETURN;
                    continue;
                }
                F.puti(depth, F_I0, ref, (int)node);
                F.puti(depth, F_I1, PT_GLASS_R, 0);
                F.puti(depth + 1, F_I0, 0, (int)(2 * node));
                L.depth = depth + 1;
                set_ray(L, point + bias, rdir, RAY_CLOSEST);
                return;
            } else {  // diffuse, :585-592
                D3 nd = hemisphere(nf, rng_u(rng, node, DIM_HEMI_1), rng_u(rng, node, DIM_HEMI_2));
                double cosTheta = std_max(0.0, dot(nd, nf));
                if (!child_possible) {
                    V = direct + mk(0, 0, 0) * (albedo / M_PI) * cosTheta;
                    pc = PC_RETURN;
                    continue;
                }
                F.at(depth, F_E + 1) = cosTheta;
                F.puti(depth, F_I0, ref, (int)node);
                F.puti(depth, F_I1, PT_DIFFUSE, 0);
                F.puti(depth + 1, F_I0, 0, (int)(2 * node));
                L.depth = depth + 1;
                set_ray(L, point + nf * A.bias, nd, RAY_CLOSEST);
                return;
            }
        } else {  // PC_RETURN: V = traceRayPath at `depth`
            if (depth == 0) {
                const int pix_slot = A.max_depth;
                D3 color = F.get3(pix_slot, 0);
                color = color + V;
                F.put3(pix_slot, 0, color);
                ++L.sample;
                pt_next_sample(A, F, L, rng);
                return;
            }
            --depth;
            int stage, unused, inode;
            F.geti(depth, F_I1, stage, unused);
            F.geti(depth, F_I0, ref, inode);
            node = (unsigned)inode;
            stage &= 0xFF;
            direct = F.get3(depth, F_B);
            const DMaterial& m = material_of(img, sv, ref);
            if (stage == PT_MIRROR) {
                V = direct + V * m.reflectivity;
            } else if (stage == PT_DIFFUSE) {
                albedo = F.get3(depth, F_A);
                double cosTheta = F.at(depth, F_E + 1);
                V = direct + V * (albedo / M_PI) * cosTheta;
            } else {
                point = F.get3(depth, F_POINT);
                normal = F.get3(depth, F_NORMAL);
                dir = F.get3(depth, F_DIR);
                D3 n = normal;
                double eta_i = 1.0, eta_t = m.ior;
                D3 incident = normalize(dir);
                if (!(dot(incident, n) < 0)) {
                    double s = eta_i; eta_i = eta_t; eta_t = s;
                    n = -n;
                }
                if (stage == PT_GLASS_R) {
                    // reflected colour is in; refraction next (:571-582)
                    D3 tdir = normalize(refract_v(incident, n, eta_t, eta_i));
                    if (length(tdir) > 0.0) {  // false for the NaN of total internal reflection
                        F.put3(depth, F_D, V);
                        F.puti(depth, F_I1, PT_GLASS_T, 0);
                        F.puti(depth + 1, F_I0, 0, (int)(2 * node + 1));
                        L.depth = depth + 1;
                        set_ray(L, point - n * A.bias, tdir, RAY_CLOSEST);
                        return;
                    }
                    V = direct + V;
                } else {  // PT_GLASS_T
                    double Fr = fresnel_f(incident, n, eta_t, eta_i);
                    D3 rc = F.get3(depth, F_D);
                    V = direct + (rc * Fr + V * (1.0 - Fr));
                }
            }
            pc = PC_RETURN;
        }
    }
}

#include "wavefront_kernels.inc"
#include "blockwave_kernels.inc"
#include "fused_kernels.inc"
#include "pathwave_kernels.inc"
#include "pathqueue_kernels.inc"

}  // namespace
