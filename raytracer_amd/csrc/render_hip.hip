// render_hip.hip -- the MI355X (gfx950) render loop behind include/raytracer_amd.h.
//
// Replaces RayTracer::render / renderPathTrace and everything below them
// (/root/reference/src/RayTracer.cpp:206-350, 373-812; Scene.cpp:63-81; BVHNode.cpp:47-81;
// BoundingBox.cpp:24-62; Sphere/Triangle/Cylinder::intersect; Camera::getRay; AreaLight::sample;
// Material::getTextureColor; toneMap).
//
// Design (DESIGN.md has the long form):
//   * One persistent kernel per frame.  Every lane of every wave repeats "rounds": trace ONE ray
//     (primary, shadow, mirror, refraction or diffuse bounce -- all of them funnel through the same
//     closest-hit walk, as they do through Scene::intersect in the reference), then run a short
//     continuation that either emits the lane's next ray or finishes the pixel and pulls a new one
//     from a global ticket counter.  So the SIMD unit of work is a ray, not a pixel: a lane stuck
//     in an 80-ray glass pixel does not idle its 63 neighbours.
//   * The recursion of traceRay/computeShadingPhong/traceRayPath is an explicit per-lane frame
//     stack in HBM, laid out [depth][field][lane] so that a wave's loads and stores coalesce.
//     Children are evaluated in the reference's order and combined with the reference's
//     expressions (post-order), so colours differ from the CPU only where pow/cos/sin/atan2/acos
//     of the device library differ from glibc in the last bits.
//   * The scene (flattened BVH + type-tagged primitive arrays) is one "image" blob that every
//     workgroup copies into LDS when it fits (all shipped scenes do: clutter_500 is 48 KB); the
//     walk then runs out of LDS with per-lane node addresses.  Larger scenes walk the same image
//     in global memory (L2-resident).
//   * The BVH walk is stackless: the reference enters both children of every node whose box is
//     hit, in a fixed order, so the visit sequence is a fixed pre-order walk with skip links.
//     Ties resolve to the LAST leaf in that order (BVHNode.cpp:64-69); the linear scan keeps the
//     FIRST object (Scene.cpp:73).
//
// Compile with -ffp-contract=off (see device_math.hpp).
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
};

// Adds a wave's tallies to the row this wave owns (rows are per (block, wave); kernels on one
// stream run one after another, so a plain read-modify-write is enough).  4096 waves doing 5
// atomics each on the same 5 words cost ~250 us per launch; this costs nothing.
// `overwrite`: the first kernel of a frame stores instead of adding, which spares the host a 2 MB memset per
// frame (a fill kernel that has to wait for a free CU when several frames are in flight).
RT_DEV void flush_tally(const Tally& tally, unsigned long long* counters, bool overwrite = false) {
    unsigned long long c[5] = {tally.rays, tally.box, tally.sph, tally.tri, tally.cyl};
    for (int off = 32; off > 0; off >>= 1)
        for (int k = 0; k < 5; ++k) c[k] += __shfl_down(c[k], off);
    if ((threadIdx.x & 63) == 0) {
        unsigned row = blockIdx.x * (blockDim.x >> 6) + (threadIdx.x >> 6);
        if (row < (unsigned)kCounterSlots) {
            unsigned long long* r = counters + (size_t)row * kCounterRow;
            if (overwrite)
                for (int k = 0; k < 5; ++k) r[k] = c[k];
            else
                for (int k = 0; k < 5; ++k) r[k] += c[k];
        } else {
            for (int k = 0; k < 5; ++k)
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
    const double reach = 4.0 * sv.scene_scale;
    bool eligible = sv.fbvh_ok && sv.regular && ray_regular(o, d) && fabs(d.x) >= 0x1p-60 && fabs(d.y) >= 0x1p-60 &&
                    fabs(d.z) >= 0x1p-60 && fabs(o.x) <= reach && fabs(o.y) <= reach && fabs(o.z) <= reach;
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
RT_DEV const DMaterial& material_of(P img, const SceneView& sv, int ref) {
    const int mi = ((const int*)(img + sv.off_matidx))[gid_of_ref(img, sv, ref)];
    // the records themselves are in the staged part only when they fit next to the geometry
    const unsigned char* base = sv.mat_staged ? img + 0 : sv.image;
    return *(const DMaterial*)(base + sv.off_mat + mi * (int)sizeof(DMaterial));
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

template <typename P>
RT_DEV D3 surface_colour(P img, const SceneView& sv, const DMaterial& m, int ref, D3 point) {
    if (m.texture < 0) return ld3(m.diffuse);
    double u, v;
    prim_uv(img, sv, ref, point, u, v);
    return texture_color(sv, m, u, v);
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
enum { DIM_RR = 0, DIM_HEMI_1 = 1, DIM_HEMI_2 = 2, DIM_LIGHT0 = 4 };

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
RT_DEV void area_sample(const RenderArgs& A, const DLight& Lt, const Rng& rng, unsigned node, int li, int si, D3 point,
                        D3& ldir, double& distance, double& pdf) {
    unsigned dim = DIM_LIGHT0 + 2u * (unsigned)(li * A.light_samples + si);
    double su = (rng_u(rng, node, dim) - 0.5) * Lt.width;
    double sv = (rng_u(rng, node, dim + 1) - 0.5) * Lt.height;
    D3 sp = ld3(Lt.position) + ld3(Lt.u) * su + ld3(Lt.v) * sv;
    ldir = sp - point;
    distance = length(ldir);
    ldir = ldir / distance;
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
    double r1 = (sx + rng_u(rng, 0, 0)) / A.sqrt_spp;
    double r2 = (sy + rng_u(rng, 0, 1)) / A.sqrt_spp;
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

// ------------------------------------------------------------------------------------------------
// The persistent kernel
// ------------------------------------------------------------------------------------------------
// kMode: 0 = Whitted / binary, 1 = path trace.  kFast: closest_hit_fast vs the literal walk.
template <int kMode, bool kLds, bool kFast>
__global__ void __launch_bounds__(kBlock, RT_WAVES_PER_EU) render_kernel(SceneView sv, RenderArgs A) {
    extern __shared__ __attribute__((aligned(16))) unsigned char lds_image[];
    const unsigned char* img;
    if (kLds) {
        // stage the scene image in LDS, 16 bytes per lane per step
        const uint4* src = (const uint4*)sv.image;
        uint4* dst = (uint4*)lds_image;
        int n16 = sv.lds_bytes >> 4;
        for (int k = threadIdx.x; k < n16; k += blockDim.x) dst[k] = src[k];
        __syncthreads();
        img = lds_image;
    } else {
        img = sv.image;
    }

    Frames F;
    F.base = A.frames;
    F.stride = A.frame_stride;
    F.tid = (long long)blockIdx.x * blockDim.x + threadIdx.x;

    Lane L;
    L.kind = RAY_NONE;
    L.exhausted = false;
    L.depth = 0;
    L.px = L.lrow = 0;
    L.sample = 0;
    L.o = L.d = mk(0, 0, 0);
    Rng rng;
    rng.k0 = A.seed_lo;
    rng.k1 = A.seed_hi;
    rng.pixel = rng.sample = 0;
    Tally tally = {0, 0, 0, 0, 0};

    for (;;) {
        // refill: lanes without a ray pull the next pixel ticket (8x8 tiles of the local rows)
        while (L.kind == RAY_NONE && !L.exhausted) {
            unsigned k = atomicAdd(A.ticket, 1u);
            if (k >= (unsigned)A.total_items) {
                L.exhausted = true;
                break;
            }
            unsigned tile = k >> 6, in = k & 63;
            int px = (int)(tile % A.tiles_x) * 8 + (int)(in & 7);
            int lrow = (int)(tile / A.tiles_x) * 8 + (int)(in >> 3);
            if (px >= A.W || lrow >= A.local_rows) continue;
            L.px = px;
            L.lrow = lrow;
            int j = global_row(A, lrow);
            if (kMode == 1) {
                rng.pixel = (unsigned)((long long)j * A.W + px);
                F.put3(A.max_depth, 0, mk(0, 0, 0));
                L.sample = A.s_begin;
                pt_next_sample(A, F, L, rng);  // may finish the pixel at once when the range is empty
            } else if (A.max_depth <= 0) {
                // traceRay(ray, 0) returns the background before intersecting (RayTracer.cpp:374-377)
                if (A.hit_id) A.hit_id[(long long)lrow * A.W + px] = -1;
                finish_pixel_whitted<const unsigned char*>(A, L, ld3(sv.bg));
            } else {  // RayTracer.cpp:217-220
                double u = 1.0 - ((double)px / (A.W - 1));
                double v = (double)j / (A.H - 1);
                D3 origin, dir;
                camera_ray(A, u, v, false, nullptr, origin, dir);
                L.depth = 0;
                set_ray(L, origin, dir, RAY_CLOSEST);
            }
        }
        if (!__any(L.kind != RAY_NONE)) break;  // every lane is out of work and out of tickets

        if (L.kind != RAY_NONE) {
            ++tally.rays;
            HitT h = closest_hit<kFast>(img, sv, L.o, L.d, tally);
            if (kMode == 1) advance_path(img, sv, A, F, L, rng, h);
            else advance_whitted(img, sv, A, F, L, h);
        }
    }

    // one atomic per wave per counter
    flush_tally(tally, A.counters);
}

// RayTracer.cpp:301-315 on a flat buffer
__global__ void resolve_kernel(double* buf, long long n_pixels, int pixel_samples, int tonemap, double exposure) {
    long long k = (long long)blockIdx.x * blockDim.x + threadIdx.x;
    if (k >= n_pixels) return;
    D3 color = mk(buf[k * 3], buf[k * 3 + 1], buf[k * 3 + 2]);
    color = color / (double)pixel_samples;
    color = tone_map(color, tonemap);
    color = color * exposure;
    color.x = pow(color.x, 1.0 / 2.2);
    color.y = pow(color.y, 1.0 / 2.2);
    color.z = pow(color.z, 1.0 / 2.2);
    buf[k * 3] = clamp01(color.x);
    buf[k * 3 + 1] = clamp01(color.y);
    buf[k * 3 + 2] = clamp01(color.z);
}

// writeImageToPPM's component conversion (RayTracer.cpp:359-362): int(255.999 * c), c already in [0, 1]
__global__ void __launch_bounds__(256) quantize_kernel(const double* rgb, long long n, unsigned char* out) {
    long long stride = (long long)gridDim.x * blockDim.x;
    for (long long i = (long long)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += stride) {
        out[i] = ppm_byte(rgb[i]);
    }
}

#include "wavefront_kernels.inc"
#include "blockwave_kernels.inc"
#include "fused_kernels.inc"
#include "pathwave_kernels.inc"
#include "gridwave_kernels.inc"

// ---- diagnostics kernels (rt_debug_*) ------------------------------------------------------------
__global__ void debug_trace_kernel(SceneView sv, int fast, long long n, const double* origins, const double* dirs,
                                   double* t_out, int* prim_out, double* normal_out) {
    long long k = (long long)blockIdx.x * blockDim.x + threadIdx.x;
    if (k >= n) return;
    const unsigned char* img = sv.image;
    D3 o = ld3(origins + 3 * k);
    D3 d = normalize(ld3(dirs + 3 * k));
    Tally tally = {0, 0, 0, 0, 0};
    HitT h = fast == 3 ? closest_hit_fbvh<false>(img, sv, o, d, tally)
           : fast == 2 ? closest_hit_fast<true, false>(img, sv, o, d, tally)
                       : (fast == 1 ? closest_hit_fast<false, false>(img, sv, o, d, tally) : closest_hit_ref(img, sv, o, d, tally));
    if (h.ref >= 0) {
        D3 point, normal;
        hit_geometry(img, sv, h, o, d, point, normal);
        t_out[k] = h.t;
        prim_out[k] = gid_of_ref(img, sv, h.ref);
        normal_out[3 * k] = normal.x;
        normal_out[3 * k + 1] = normal.y;
        normal_out[3 * k + 2] = normal.z;
    } else {
        t_out[k] = -1.0;
        prim_out[k] = -1;
        normal_out[3 * k] = normal_out[3 * k + 1] = normal_out[3 * k + 2] = 0.0;
    }
}

__global__ void debug_math_kernel(int op, long long n, const double* a, const double* b, double* out) {
    long long k = (long long)blockIdx.x * blockDim.x + threadIdx.x;
    if (k >= n) return;
    double x = a[k], y = b ? b[k] : 0.0, r;
    switch (op) {
        case 0: r = x / y; break;
        case 1: r = sqrt(x); break;
        case 2: r = pow(x, y); break;
        case 3: r = atan2(x, y); break;
        case 4: r = acos(x); break;
        case 5: r = cos(x); break;
        case 6: r = sin(x); break;
        case 7: {  // uniform(seed = 0; pixel = x, sample = y, node 1, dim 0) for RNG parity
            Rng g;
            g.k0 = 0; g.k1 = 0;
            g.pixel = (unsigned)x;
            g.sample = (unsigned)y;
            r = rng_u(g, 1, 0);
            break;
        }
        case 8: r = div_shared(x, y, refine_rcp(y)); break;  // exact division through a shared reciprocal
        case 9: r = normalize(mk(x, y, 0.75 * y)).x; break;  // Vector3::normalize (shared reciprocal when it is exact)
        case 10: r = normalize(mk(y, 0.75 * y, x)).z; break;
        default: r = 0.0;
    }
    out[k] = r;
}

}  // namespace

// ------------------------------------------------------------------------------------------------
// Host side of the device library
// ------------------------------------------------------------------------------------------------
#define HIP_TRY(expr)                                                                                      \
    do {                                                                                                   \
        hipError_t e_ = (expr);                                                                            \
        if (e_ != hipSuccess) return rt::set_error(RT_ERR_DEVICE, "%s -> %s", #expr, hipGetErrorString(e_)); \
    } while (0)

struct rt_context {
    int device = 0;
    int n_cus = 0;
    std::map<std::tuple<const void*, int, size_t>, int> grid_cache;  // wf_persistent_grid: workgroups per CU
    int max_workgroups = 0;  // cap on a persistent kernel's grid (0: as many as the chip holds), RT_WORKGROUPS / rt_set_workgroups
    bool have_scene = false, have_camera = false, have_params = false;
    SceneView sv;
    rt_camera cam;
    rt_params params;
    // device allocations
    unsigned char* d_image = nullptr;
    DMaterial* d_materials = nullptr;
    int* d_mat_of_gid = nullptr;
    DLight* d_lights = nullptr;
    DTexture* d_textures = nullptr;
    unsigned char* d_texels = nullptr;
    double* d_frames = nullptr;
    size_t frames_bytes = 0;
    unsigned int* d_ticket = nullptr;
    unsigned long long* d_counters = nullptr;
    hipEvent_t ev_start = nullptr, ev_stop = nullptr;
    bool timed = false;
    hipStream_t last_stream = nullptr;
    unsigned long long self_clean_gen = 0;  // generation whose last frame left counters / hand-over counts ready for the next (blockwave_host.inc)
    hipStream_t self_clean_stream = nullptr;
    double upload_ms = 0.0;
    int n_bvh_nodes = 0;
    int scene_in_lds = 0;
    int sqrt_spp = 0;
    bool fast_traversal = true;  // false: RT_TRAVERSAL=ref, the literal walk with true divisions
    int traversal = 3;           // 0 ref, 1 walk (same visits as the reference), 2 cull (+ any-hit shadows), 3 filter BVH
    bool wavefront = true;       // RT_PIPELINE=mega selects the persistent per-pixel kernel
    bool level_kernels = false;  // RT_PIPELINE=levels: one kernel per phase and depth instead of one per frame
    bool pt_generic = false;     // RT_PT_GENERIC=1: path trace with bw_kernel<1> (any lights / sample counts) even where pt_kernel applies
    int fused = -1;              // one pass per level (fused_kernels.inc): -1 = Whitted yes, path trace no; RT_PIPELINE=fused / block force it
    unsigned* d_bw_info = nullptr;
    uint2* d_gw_lvl = nullptr;       // gridwave: per level, per workgroup (first id, count)
    unsigned* d_gw_barrier = nullptr;
    long long* d_gw_timing = nullptr;
    int gw_lvl_blocks = 0;
    bool gridwave = false;           // RT_PIPELINE=grid: per-level dealing over the whole grid (gridwave_kernels.inc)
    bool split_roots = true;         // fused pipeline: level 0 in fw_roots_kernel (RT_SPLIT_ROOTS=0: inside the frame kernel)
    unsigned* d_k1_count = nullptr;
    int k1_count_blocks = 0;
    int bw_info_blocks = 0;
    unsigned long long generation = 1;  // bumped by upload / set_camera / set_params
    int n_point_lights = 0, n_area_lights = 0;
    // wavefront node store
    double* d_wf_nodes = nullptr;
    int* d_wf_alive = nullptr;
    unsigned char* d_wf_shadow = nullptr;
    void* d_wf_state = nullptr;
    long long wf_cap = 0;
    size_t wf_shadow_bytes = 0;
    unsigned long long wf_calibrated_gen = 0;
    long long wf_needed_cap = 0;
    int wf_batch = 1, wf_levels_used = 0;
    bool last_was_wavefront = false;
};

namespace {

void free_scene(rt_context* c) {
    (void)hipFree(c->d_image);
    (void)hipFree(c->d_materials);
    (void)hipFree(c->d_mat_of_gid);
    (void)hipFree(c->d_lights);
    (void)hipFree(c->d_textures);
    (void)hipFree(c->d_texels);
    c->d_image = nullptr;
    c->d_materials = nullptr;
    c->d_mat_of_gid = nullptr;
    c->d_lights = nullptr;
    c->d_textures = nullptr;
    c->d_texels = nullptr;
    c->have_scene = false;
}

int64_t local_rows_of(const rt_params& p) {
    int br = p.band_rows > 0 ? p.band_rows : 8;
    int world = p.band_world > 0 ? p.band_world : 1;
    int64_t n_bands = (p.height + br - 1) / br;
    int64_t rows = 0;
    for (int64_t b = p.band_rank; b < n_bands; b += world) {
        int64_t r0 = b * br, r1 = std::min<int64_t>(p.height, r0 + br);
        rows += r1 - r0;
    }
    return rows;
}

int align16(int x) { return (x + 15) & ~15; }

struct H3 {
    double x, y, z;
};
H3 h_normalize(const double* p) {  // Vector3::normalize on the host, same operation order
    double len = std::sqrt(p[0] * p[0] + p[1] * p[1] + p[2] * p[2]);
    return H3{p[0] / len, p[1] / len, p[2] / len};
}

template <typename T>
int upload(T** dst, const std::vector<T>& src) {
    size_t bytes = std::max<size_t>(src.size(), 1) * sizeof(T);
    HIP_TRY(hipMalloc((void**)dst, bytes));
    if (!src.empty()) HIP_TRY(hipMemcpy(*dst, src.data(), src.size() * sizeof(T), hipMemcpyHostToDevice));
    return RT_OK;
}

#include "wavefront_host.inc"
#include "blockwave_host.inc"
#include "gridwave_host.inc"

int launch_render(rt_context* c, hipStream_t stream, double* rgb_dev, int* hit_dev, bool accumulate_only,
                  unsigned char* rgb8_dev = nullptr) {
    if (!c->have_scene || !c->have_camera || !c->have_params)
        return rt::set_error(RT_ERR_STATE, "render: scene, camera and params must be set first");
    const rt_params& p = c->params;
    HIP_TRY(hipSetDevice(c->device));
    RenderArgs A;
    std::memset(&A, 0, sizeof A);
    std::memcpy(A.cam_pos, c->cam.position, sizeof A.cam_pos);
    std::memcpy(A.cam_ll, c->cam.lower_left, sizeof A.cam_ll);
    std::memcpy(A.cam_hor, c->cam.horizontal, sizeof A.cam_hor);
    std::memcpy(A.cam_ver, c->cam.vertical, sizeof A.cam_ver);
    std::memcpy(A.cam_u, c->cam.u, sizeof A.cam_u);
    std::memcpy(A.cam_v, c->cam.v, sizeof A.cam_v);
    A.lens_radius = c->cam.lens_radius;
    A.exposure = p.exposure;
    A.bias = p.shadow_bias;
    A.W = p.width;
    A.H = p.height;
    A.mode = p.mode;
    A.tonemap = p.tonemap;
    A.max_depth = p.max_depth;
    A.pixel_samples = p.pixel_samples;
    A.light_samples = p.light_samples;
    A.sqrt_spp = c->sqrt_spp;
    A.s_begin = std::max(0, p.sample_begin);
    A.s_end = p.sample_end > 0 ? std::min(p.sample_end, c->sqrt_spp * c->sqrt_spp) : c->sqrt_spp * c->sqrt_spp;
    A.band_rows = p.band_rows;
    A.band_rank = p.band_rank;
    A.band_world = p.band_world;
    A.local_rows = (int)local_rows_of(p);
    A.tiles_x = (p.width + 7) / 8;
    int tiles_y = (A.local_rows + 7) / 8;
    long long total = (long long)A.tiles_x * tiles_y * 64;
    if (total >= 0x7FFFFFFFll) return rt::set_error(RT_ERR_LIMIT, "render: more than 2^31 pixel tickets in one call");
    A.total_items = (int)total;
    A.seed_lo = (unsigned)(p.seed & 0xFFFFFFFFull);
    A.seed_hi = (unsigned)(p.seed >> 32);
    A.accumulate_only = accumulate_only ? 1 : 0;
    A.rgb = rgb_dev;
    A.rgb8 = rgb8_dev;
    A.hit_id = hit_dev;
    A.ticket = c->d_ticket;
    A.counters = c->d_counters;

    c->last_was_wavefront = c->wavefront;
    if (c->wavefront) {
        c->scene_in_lds = c->sv.lds_bytes <= kMaxLdsImage ? 1 : 0;
        if (c->level_kernels) return launch_wavefront(c, stream, A, accumulate_only);
        if (c->gridwave && p.mode != RT_MODE_PATHTRACE && p.max_depth > 0) return launch_gridwave(c, stream, A);
        return launch_blockwave(c, stream, A, accumulate_only);
    }

    // persistent grid: as many workgroups as the chip holds at once
    bool lds = c->sv.lds_bytes <= kMaxLdsImage;
    size_t lds_bytes = lds ? (size_t)c->sv.lds_bytes : 0;
    const int kmode = p.mode == RT_MODE_PATHTRACE ? 1 : 0;
    using KernelFn = void (*)(SceneView, RenderArgs);
    static const KernelFn table[2][2][2] = {
        {{render_kernel<0, false, false>, render_kernel<0, false, true>}, {render_kernel<0, true, false>, render_kernel<0, true, true>}},
        {{render_kernel<1, false, false>, render_kernel<1, false, true>}, {render_kernel<1, true, false>, render_kernel<1, true, true>}}};
    KernelFn kernel = table[kmode][lds ? 1 : 0][c->fast_traversal ? 1 : 0];
    if (lds_bytes > 48 * 1024)
        HIP_TRY(hipFuncSetAttribute((const void*)kernel, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds_bytes));
    int per_cu = 0;
    HIP_TRY(hipOccupancyMaxActiveBlocksPerMultiprocessor(&per_cu, (const void*)kernel, kBlock, lds_bytes));
    if (per_cu < 1) per_cu = 1;
    long long want_blocks = (total + kBlock - 1) / kBlock;
    int grid = (int)std::min<long long>((long long)per_cu * c->n_cus, std::max<long long>(want_blocks, 1));
    c->scene_in_lds = lds ? 1 : 0;

    long long n_threads = (long long)grid * kBlock;
    size_t need = (size_t)(p.max_depth + 1) * kNF * (size_t)n_threads * sizeof(double);
    if (need > c->frames_bytes) {
        if (c->d_frames) {
            HIP_TRY(hipStreamSynchronize(stream));
            (void)hipFree(c->d_frames);
            c->d_frames = nullptr;
            c->frames_bytes = 0;
        }
        HIP_TRY(hipMalloc((void**)&c->d_frames, need));
        c->frames_bytes = need;
    }
    A.frames = c->d_frames;
    A.frame_stride = n_threads;

    HIP_TRY(hipMemsetAsync(c->d_ticket, 0, sizeof(unsigned int), stream));
    HIP_TRY(hipMemsetAsync(c->d_counters, 0, (size_t)kCounterSlots * kCounterRow * sizeof(unsigned long long), stream));
    HIP_TRY(hipEventRecord(c->ev_start, stream));
    hipLaunchKernelGGL(kernel, dim3(grid), dim3(kBlock), lds_bytes, stream, c->sv, A);
    HIP_TRY(hipGetLastError());
    HIP_TRY(hipEventRecord(c->ev_stop, stream));
    c->timed = true;
    c->last_stream = stream;
    return RT_OK;
}

}  // namespace

extern "C" {

int rt_device_count(void) {
    int n = 0;
    if (hipGetDeviceCount(&n) != hipSuccess) return 0;
    return n;
}

int rt_create(rt_context** out, int device) {
    if (!out) return rt::set_error(RT_ERR_INVALID, "rt_create: null out pointer");
    *out = nullptr;
    int n = 0;
    hipError_t e = hipGetDeviceCount(&n);
    if (e != hipSuccess || n <= 0)
        return rt::set_error(RT_ERR_DEVICE, "rt_create: no HIP device (%s); this library has no CPU fallback",
                             e == hipSuccess ? "count = 0" : hipGetErrorString(e));
    if (device < 0 || device >= n) return rt::set_error(RT_ERR_INVALID, "rt_create: device %d of %d", device, n);
    HIP_TRY(hipSetDevice(device));
    hipDeviceProp_t prop;
    HIP_TRY(hipGetDeviceProperties(&prop, device));
    if (std::strncmp(prop.gcnArchName, "gfx950", 6) != 0)
        return rt::set_error(RT_ERR_DEVICE, "rt_create: device %d is %s; the kernels are built for gfx950 only", device,
                             prop.gcnArchName);
    rt_context* c = new rt_context();
    c->device = device;
    const char* trav = std::getenv("RT_TRAVERSAL");
    c->fast_traversal = !(trav && std::strcmp(trav, "ref") == 0);
    c->traversal = !c->fast_traversal ? 0 : ((trav && std::strcmp(trav, "walk") == 0) ? 1 : ((trav && std::strcmp(trav, "cull") == 0) ? 2 : 3));
    const char* pipe = std::getenv("RT_PIPELINE");
    c->wavefront = !(pipe && std::strcmp(pipe, "mega") == 0) && c->fast_traversal;
    c->level_kernels = pipe && std::strcmp(pipe, "levels") == 0;
    c->fused = pipe && std::strcmp(pipe, "fused") == 0 ? 1 : (pipe && std::strcmp(pipe, "block") == 0 ? 0 : -1);
    c->gridwave = pipe && std::strcmp(pipe, "grid") == 0;
    if (const char* e = std::getenv("RT_SPLIT_ROOTS")) c->split_roots = std::atoi(e) != 0;
    if (const char* e = std::getenv("RT_PT_GENERIC")) c->pt_generic = std::atoi(e) != 0;
    c->n_cus = prop.multiProcessorCount;
    if (const char* e = std::getenv("RT_WORKGROUPS")) c->max_workgroups = std::max(0, std::atoi(e));
    std::memset(&c->sv, 0, sizeof c->sv);
    hipError_t a = hipMalloc((void**)&c->d_ticket, sizeof(unsigned int));
    hipError_t b = hipMalloc((void**)&c->d_counters, (size_t)kCounterSlots * kCounterRow * sizeof(unsigned long long));
    hipError_t s0 = hipEventCreate(&c->ev_start);
    hipError_t s1 = hipEventCreate(&c->ev_stop);
    if (a != hipSuccess || b != hipSuccess || s0 != hipSuccess || s1 != hipSuccess) {
        rt_destroy(c);
        return rt::set_error(RT_ERR_DEVICE, "rt_create: device allocation failed");
    }
    *out = c;
    return RT_OK;
}

void rt_destroy(rt_context* c) {
    if (!c) return;
    (void)hipSetDevice(c->device);
    (void)hipDeviceSynchronize();
    free_scene(c);
    (void)hipFree(c->d_frames);
    (void)hipFree(c->d_wf_nodes);
    (void)hipFree(c->d_wf_alive);
    (void)hipFree(c->d_wf_shadow);
    (void)hipFree(c->d_wf_state);
    (void)hipFree(c->d_bw_info);
    (void)hipFree(c->d_gw_lvl);
    (void)hipFree(c->d_gw_barrier);
    (void)hipFree(c->d_k1_count);
    (void)hipFree(c->d_gw_timing);
    (void)hipFree(c->d_ticket);
    (void)hipFree(c->d_counters);
    if (c->ev_start) (void)hipEventDestroy(c->ev_start);
    if (c->ev_stop) (void)hipEventDestroy(c->ev_stop);
    delete c;
}

int rt_upload_scene(rt_context* c, const rt_scene_desc* s) {
    if (!c || !s) return rt::set_error(RT_ERR_INVALID, "rt_upload_scene: null argument");
    if (s->n_prims < 0 || s->n_prims >= (1 << 28)) return rt::set_error(RT_ERR_LIMIT, "rt_upload_scene: n_prims out of range");
    for (int i = 0; i < s->n_prims; ++i) {
        int t = s->prim_type[i], m = s->prim_material[i];
        if (t < RT_SPHERE || t > RT_CYLINDER) return rt::set_error(RT_ERR_INVALID, "primitive %d: unknown type %d", i, t);
        if (m < 0 || m >= s->n_materials) return rt::set_error(RT_ERR_INVALID, "primitive %d: material %d of %d", i, m, s->n_materials);
    }
    for (int i = 0; i < s->n_materials; ++i) {
        int t = s->materials[i].texture;
        if (t >= s->n_textures) return rt::set_error(RT_ERR_INVALID, "material %d: texture %d of %d", i, t, s->n_textures);
    }
    auto t0 = std::chrono::steady_clock::now();
    HIP_TRY(hipSetDevice(c->device));
    HIP_TRY(hipDeviceSynchronize());
    free_scene(c);

    // ---- flatten: per-type arrays in primitive order ------------------------------------------
    std::vector<double> sph, tri, cyl;
    std::vector<int> gid_sph, gid_tri, gid_cyl, ref_of_gid(s->n_prims);
    for (int i = 0; i < s->n_prims; ++i) {
        const double* g = s->prim_geom + (size_t)i * RT_GEOM_STRIDE;
        if (s->prim_type[i] == RT_SPHERE) {
            ref_of_gid[i] = make_ref(RT_SPHERE, (int)gid_sph.size());
            gid_sph.push_back(i);
            sph.insert(sph.end(), g, g + 4);
        } else if (s->prim_type[i] == RT_TRIANGLE) {
            ref_of_gid[i] = make_ref(RT_TRIANGLE, (int)gid_tri.size());
            gid_tri.push_back(i);
            double rec[kTriDoubles];
            for (int k = 0; k < 3; ++k) {
                rec[k] = g[k];
                rec[3 + k] = g[3 + k] - g[k];  // edge1 = v1 - v0 (Triangle.cpp:44)
                rec[6 + k] = g[6 + k] - g[k];  // edge2 = v2 - v0 (Triangle.cpp:45)
            }
            rt::triangle_normal(g, g + 3, g + 6, rec + 9);
            tri.insert(tri.end(), rec, rec + kTriDoubles);
        } else {
            ref_of_gid[i] = make_ref(RT_CYLINDER, (int)gid_cyl.size());
            gid_cyl.push_back(i);
            double rec[kCylDoubles];
            H3 ax = h_normalize(g + 3);  // Cylinder ctor normalizes the axis again (Cylinder.cpp:11)
            rec[0] = g[0]; rec[1] = g[1]; rec[2] = g[2];
            rec[3] = ax.x; rec[4] = ax.y; rec[5] = ax.z;
            rec[6] = g[6];
            rec[7] = g[7];
            cyl.insert(cyl.end(), rec, rec + kCylDoubles);
        }
    }
    rt::FlatBVH bvh;
    if (s->use_bvh) rt::build_flat_bvh(s, bvh);

    SceneView sv;
    std::memset(&sv, 0, sizeof sv);
    sv.n_nodes = bvh.size();
    sv.n_prims = s->n_prims;
    {
        auto in_window = [](double x) {
            double ax = std::fabs(x);
            return ax == 0.0 || (ax >= 0x1p-300 && ax <= 0x1p300);
        };
        bool regular = true;
        for (double v : bvh.box) regular = regular && in_window(v);
        for (double v : sph) regular = regular && in_window(v);
        sv.regular = regular ? 1 : 0;
    }
    rt::FilterBVH fb;
    rt::build_filter_bvh(s, s->use_bvh ? &bvh : nullptr, fb);
    // scale cap: b * (1/d) with |1/d| <= 2^60 must stay finite in fp32 (closest_hit_fbvh)
    if (!(fb.scene_scale <= 0x1p40)) fb.ok = false;
    sv.fbvh_ok = fb.ok ? 1 : 0;
    sv.n_fnodes = fb.n_nodes;
    sv.scene_scale = fb.scene_scale;
    // Image layout: first what every ray of the filter-BVH traversal touches (staged into LDS), then the
    // reference tree and the linear-scan table, which that traversal only reads for the rare rays it
    // hands to the exact walk (from the global copy).  The other traversal modes stage everything.
    int off = 0;
    if (fb.ok) {
        sv.off_fnodes = off;   off = align16(off + fb.n_nodes * kFNodeBytes);
        sv.off_leafprim = off; off = align16(off + s->n_prims * kLeafBytes);
    }
    sv.off_sph = off;   off = align16(off + (int)sph.size() * 8);
    sv.off_tri = off;   off = align16(off + (int)tri.size() * 8);
    sv.off_cyl = off;   off = align16(off + (int)cyl.size() * 8);
    sv.off_gid_sph = off; off = align16(off + (int)gid_sph.size() * 4);
    sv.off_gid_tri = off; off = align16(off + (int)gid_tri.size() * 4);
    sv.off_gid_cyl = off; off = align16(off + (int)gid_cyl.size() * 4);
    sv.off_matidx = off;  off = align16(off + s->n_prims * 4);
    const int mat_bytes = align16(s->n_materials * (int)sizeof(DMaterial));
    sv.mat_staged = off + mat_bytes <= kMaxLdsImage ? 1 : 0;  // clutter_500: 500 materials = 48 KB do not fit
    if (sv.mat_staged) {
        sv.off_mat = off;
        off += mat_bytes;
    }
    sv.lds_bytes = std::max(off, 16);
    if (!sv.mat_staged) {
        sv.off_mat = off;
        off += mat_bytes;
    }
    sv.off_nodes = off; off = align16(off + sv.n_nodes * kNodeBytes);
    sv.off_ref = off;   off = align16(off + s->n_prims * 4);
    sv.image_bytes = std::max(off, 16);
    if (std::getenv("RT_WF_DEBUG"))
        std::fprintf(stderr, "[scene] %d prims, %d filter nodes (%d B), leaf records %d B, primitives %d B, materials %d B (%s), staged %d B of %d B\n",
                     s->n_prims, fb.n_nodes, fb.n_nodes * kFNodeBytes, s->n_prims * kLeafBytes, sv.off_gid_sph - sv.off_sph, mat_bytes,
                     sv.mat_staged ? "staged" : "global", sv.lds_bytes, sv.image_bytes);
    if (!(fb.ok && c->traversal == 3 && c->wavefront)) sv.lds_bytes = sv.image_bytes;  // the per-pixel kernel walks the reference tree
    // too large for LDS: the leading filter nodes are staged instead (the launch trims this to the kernel's room)
    sv.top_bytes = (fb.ok && c->traversal == 3 && c->wavefront && sv.lds_bytes > kMaxLdsImage && sv.off_fnodes == 0)
                       ? (int)std::min<long long>((long long)fb.n_nodes * kFNodeBytes, 160 * 1024) : 0;
    if (const char* e = std::getenv("RT_STAGE_TOP")) if (std::atoi(e) == 0) sv.top_bytes = 0;  // A/B: everything from the global copy
    // Culling margin per node (closest_hit_fast<kCull>): absolute slack of 1e-6 of the box's size and
    // position, or +inf when some primitive below the node is not contained in its reference box
    // (cylinders whose axis is not exactly +Y, src/Cylinder.cpp:131-137): such nodes are never culled.
    std::vector<float> cull_margin((size_t)sv.n_nodes, 0.f);
    {
        std::vector<char> loose((size_t)sv.n_nodes, 0);
        auto prim_loose = [&](int gid) {
            if (gid < 0 || s->prim_type[gid] != RT_CYLINDER) return false;
            const double* g = s->prim_geom + (size_t)gid * RT_GEOM_STRIDE;
            H3 ax = h_normalize(g + 3);
            return !(ax.x == 0.0 && ax.y == 1.0 && ax.z == 0.0);
        };
        for (int i = sv.n_nodes - 1; i >= 0; --i) {  // children follow their parent in pre-order
            bool l = prim_loose(bvh.prim0[i]) || prim_loose(bvh.prim1[i]);
            if (bvh.right[i] >= 0) l = l || loose[i + 1] || loose[bvh.right[i]];
            loose[i] = l ? 1 : 0;
            const double* b = &bvh.box[(size_t)i * 6];
            double scale = 0.0;
            for (int k = 0; k < 6; ++k) scale = std::max(scale, std::fabs(b[k]));
            for (int k = 0; k < 3; ++k) scale += std::fabs(b[3 + k] - b[k]);
            cull_margin[i] = l ? INFINITY : (float)(1e-6 * scale + 1e-9);
        }
    }
    std::vector<unsigned char> image((size_t)sv.image_bytes, 0);
    for (int i = 0; i < sv.n_nodes; ++i) {
        unsigned char* n = image.data() + sv.off_nodes + (size_t)i * kNodeBytes;
        std::memcpy(n, &bvh.box[(size_t)i * 6], 48);
        float margin = cull_margin[i];
        int margin_bits;
        std::memcpy(&margin_bits, &margin, 4);
        int ints[4] = {bvh.skip[i], bvh.prim0[i] >= 0 ? ref_of_gid[bvh.prim0[i]] : -1,
                       bvh.prim1[i] >= 0 ? ref_of_gid[bvh.prim1[i]] : -1, margin_bits};
        std::memcpy(n + 48, ints, 16);
    }
    if (!sph.empty()) std::memcpy(image.data() + sv.off_sph, sph.data(), sph.size() * 8);
    if (!tri.empty()) std::memcpy(image.data() + sv.off_tri, tri.data(), tri.size() * 8);
    if (!cyl.empty()) std::memcpy(image.data() + sv.off_cyl, cyl.data(), cyl.size() * 8);
    if (!gid_sph.empty()) std::memcpy(image.data() + sv.off_gid_sph, gid_sph.data(), gid_sph.size() * 4);
    if (!gid_tri.empty()) std::memcpy(image.data() + sv.off_gid_tri, gid_tri.data(), gid_tri.size() * 4);
    if (!gid_cyl.empty()) std::memcpy(image.data() + sv.off_gid_cyl, gid_cyl.data(), gid_cyl.size() * 4);
    if (s->n_prims) std::memcpy(image.data() + sv.off_ref, ref_of_gid.data(), (size_t)s->n_prims * 4);
    if (fb.ok) {
        for (int i = 0; i < fb.n_nodes; ++i) {
            unsigned char* n = image.data() + sv.off_fnodes + (size_t)i * kFNodeBytes;
            const float* bx = &fb.box[(size_t)i * 6];  // lo xyz, hi xyz -> (lo, hi) pairs per axis
            const float pairs[6] = {bx[0], bx[3], bx[1], bx[4], bx[2], bx[5]};
            std::memcpy(n, pairs, 24);
            std::memcpy(n + 24, &fb.leaf_first[i], 4);
            std::memcpy(n + 28, &fb.leaf_count[i], 4);
            int links[16];  // node indices -> byte offsets in the image, so that a walk step is one add
            for (int k = 0; k < 16; ++k) {
                int to = fb.links[(size_t)i * 16 + k];
                links[k] = to < 0 ? -1 : sv.off_fnodes + to * kFNodeBytes;
                // a leaf's "pass" link names its records instead (closest_hit_fbvh); the walk then goes on
                // along the "fail" link, which for a leaf is the same node
                if (fb.leaf_count[i] > 0 && (k & 1) == 0) links[k] = -2 - ((fb.leaf_first[i] << 2) | fb.leaf_count[i]);
            }
            std::memcpy(n + 32, links, 64);
        }
        for (int k = 0; k < s->n_prims; ++k) {
            const int gate = fb.leaf_prim[(size_t)k * 4 + 1];  // reference node whose box gates the primitive, or -1
            int rec[4] = {ref_of_gid[fb.leaf_prim[(size_t)k * 4]], gate, fb.leaf_prim[(size_t)k * 4 + 2], 0};
            unsigned char* dst = image.data() + sv.off_leafprim + (size_t)k * kLeafBytes;
            std::memcpy(dst, rec, 16);
            if (gate >= 0) std::memcpy(dst + 16, &bvh.box[(size_t)gate * 6], 48);
        }
    }

    std::vector<DMaterial> mats(s->n_materials);
    for (int i = 0; i < s->n_materials; ++i) {
        const rt_material& m = s->materials[i];
        DMaterial& d = mats[i];
        d.ks = m.ks; d.kd = m.kd; d.reflectivity = m.reflectivity; d.ior = m.refractive_index;
        std::memcpy(d.diffuse, m.diffuse, sizeof d.diffuse);
        std::memcpy(d.specular, m.specular, sizeof d.specular);
        d.spec_exp = m.specular_exponent;
        d.reflective = m.is_reflective;
        d.refractive = m.is_refractive;
        d.texture = m.texture;
    }
    std::vector<int> mat_of_gid(s->prim_material, s->prim_material + s->n_prims);
    std::vector<DLight> lights(s->n_lights);
    for (int i = 0; i < s->n_lights; ++i) {
        const rt_light& l = s->lights[i];
        DLight& d = lights[i];
        std::memset(&d, 0, sizeof d);
        d.type = l.type;
        std::memcpy(d.intensity, l.intensity, sizeof d.intensity);
        std::memcpy(d.position, l.position, sizeof d.position);
        if (l.type == RT_LIGHT_AREA) {  // AreaLight ctor normalizes (include/AreaLight.h:21-22)
            H3 n = h_normalize(l.normal), u = h_normalize(l.u), v = h_normalize(l.v);
            d.normal[0] = n.x; d.normal[1] = n.y; d.normal[2] = n.z;
            d.u[0] = u.x; d.u[1] = u.y; d.u[2] = u.z;
            d.v[0] = v.x; d.v[1] = v.y; d.v[2] = v.z;
        }
        d.width = l.width;
        d.height = l.height;
    }
    std::vector<DTexture> texs(s->n_textures);
    std::vector<unsigned char> texels;
    for (int i = 0; i < s->n_textures; ++i) {
        const rt_texture& t = s->textures[i];
        texs[i].width = t.width;
        texs[i].height = t.height;
        texs[i].offset = (long long)texels.size();
        size_t n = (t.width > 0 && t.height > 0 && t.rgb) ? (size_t)t.width * t.height * 3 : 0;
        if (n == 0) texs[i].width = texs[i].height = 0;
        else texels.insert(texels.end(), t.rgb, t.rgb + n);
    }

    if (s->n_prims) std::memcpy(image.data() + sv.off_matidx, mat_of_gid.data(), (size_t)s->n_prims * 4);
    if (s->n_materials) std::memcpy(image.data() + sv.off_mat, mats.data(), (size_t)s->n_materials * sizeof(DMaterial));
    int rc;
    if ((rc = upload(&c->d_image, image)) != RT_OK) return rc;
    if ((rc = upload(&c->d_materials, mats)) != RT_OK) return rc;
    if ((rc = upload(&c->d_mat_of_gid, mat_of_gid)) != RT_OK) return rc;
    if ((rc = upload(&c->d_lights, lights)) != RT_OK) return rc;
    if ((rc = upload(&c->d_textures, texs)) != RT_OK) return rc;
    if ((rc = upload(&c->d_texels, texels)) != RT_OK) return rc;
    sv.image = c->d_image;
    sv.materials = c->d_materials;
    sv.mat_of_gid = c->d_mat_of_gid;
    sv.lights = c->d_lights;
    sv.n_lights = s->n_lights;
    sv.textures = c->d_textures;
    sv.texels = c->d_texels;
    std::memcpy(sv.bg, s->background, sizeof sv.bg);
    c->sv = sv;
    c->n_bvh_nodes = sv.n_nodes;
    c->n_point_lights = c->n_area_lights = 0;
    for (int i = 0; i < s->n_lights; ++i) (s->lights[i].type == RT_LIGHT_POINT ? c->n_point_lights : c->n_area_lights)++;
    ++c->generation;
    c->have_scene = true;
    c->upload_ms = std::chrono::duration<double, std::milli>(std::chrono::steady_clock::now() - t0).count();
    return RT_OK;
}

int rt_set_camera(rt_context* c, const rt_camera* cam) {
    if (!c || !cam) return rt::set_error(RT_ERR_INVALID, "rt_set_camera: null argument");
    c->cam = *cam;
    ++c->generation;
    c->have_camera = true;
    return RT_OK;
}

int rt_set_params(rt_context* c, const rt_params* p) {
    if (!c || !p) return rt::set_error(RT_ERR_INVALID, "rt_set_params: null argument");
    if (p->width < 2 || p->height < 2)
        return rt::set_error(RT_ERR_INVALID, "rt_set_params: width and height must be >= 2 (u = i/(W-1))");
    if (p->mode < RT_MODE_PHONG || p->mode > RT_MODE_PATHTRACE) return rt::set_error(RT_ERR_INVALID, "rt_set_params: mode %d", p->mode);
    if (p->tonemap < RT_TONEMAP_NONE || p->tonemap > RT_TONEMAP_UNCHARTED2)
        return rt::set_error(RT_ERR_INVALID, "rt_set_params: tonemap %d", p->tonemap);
    if (p->max_depth < 0 || p->max_depth > kMaxDepthLimit)
        return rt::set_error(RT_ERR_LIMIT, "rt_set_params: max_depth %d outside [0, %d]", p->max_depth, kMaxDepthLimit);
    if (p->mode == RT_MODE_PATHTRACE && p->max_depth > 30)
        return rt::set_error(RT_ERR_LIMIT, "rt_set_params: path trace max_depth %d > 30 (RNG node code is 32 bits)", p->max_depth);
    if (p->band_rows <= 0 || p->band_world <= 0 || p->band_rank < 0 || p->band_rank >= p->band_world)
        return rt::set_error(RT_ERR_INVALID, "rt_set_params: band partition %d/%d rows %d", p->band_rank, p->band_world, p->band_rows);
    c->params = *p;
    ++c->generation;
    c->sqrt_spp = 0;
    if (p->mode == RT_MODE_PATHTRACE) {
        if (p->pixel_samples < 0) return rt::set_error(RT_ERR_INVALID, "rt_set_params: pixel_samples %d", p->pixel_samples);
        c->sqrt_spp = static_cast<int>(std::sqrt((double)p->pixel_samples));  // RayTracer.cpp:255
    }
    c->have_params = true;
    return RT_OK;
}

int64_t rt_local_rows(const rt_context* c) {
    if (!c || !c->have_params) return rt::set_error(RT_ERR_STATE, "rt_local_rows: params not set");
    return local_rows_of(c->params);
}

int rt_render_device(rt_context* c, void* stream, double* rgb_dev, int32_t* hit_id_dev) {
    if (!c || !rgb_dev) return rt::set_error(RT_ERR_INVALID, "rt_render_device: null argument");
    return launch_render(c, (hipStream_t)stream, rgb_dev, hit_id_dev, false);
}

int rt_pathtrace_accumulate_device(rt_context* c, void* stream, double* accum_dev) {
    if (!c || !accum_dev) return rt::set_error(RT_ERR_INVALID, "rt_pathtrace_accumulate_device: null argument");
    if (!c->have_params || c->params.mode != RT_MODE_PATHTRACE)
        return rt::set_error(RT_ERR_STATE, "rt_pathtrace_accumulate_device: mode is not path trace");
    return launch_render(c, (hipStream_t)stream, accum_dev, nullptr, true);
}

int rt_pathtrace_resolve_device(rt_context* c, void* stream, double* accum_dev, int64_t n_pixels) {
    if (!c || !accum_dev || n_pixels < 0) return rt::set_error(RT_ERR_INVALID, "rt_pathtrace_resolve_device: bad argument");
    if (!c->have_params) return rt::set_error(RT_ERR_STATE, "rt_pathtrace_resolve_device: params not set");
    if (n_pixels == 0) return RT_OK;
    HIP_TRY(hipSetDevice(c->device));
    int block = 256;
    long long grid = (n_pixels + block - 1) / block;
    hipLaunchKernelGGL(resolve_kernel, dim3((unsigned)grid), dim3(block), 0, (hipStream_t)stream, accum_dev, (long long)n_pixels,
                       c->params.pixel_samples, c->params.tonemap, c->params.exposure);
    HIP_TRY(hipGetLastError());
    return RT_OK;
}

int rt_render_device_u8(rt_context* c, void* stream, uint8_t* rgb8_dev, double* rgb_dev, int32_t* hit_id_dev) {
    if (!c || !rgb8_dev) return rt::set_error(RT_ERR_INVALID, "rt_render_device_u8: null argument");
    if (!c->have_params) return rt::set_error(RT_ERR_STATE, "rt_render_device_u8: params not set");
    if (c->params.mode == RT_MODE_PATHTRACE)
        return rt::set_error(RT_ERR_INVALID, "rt_render_device_u8: path-trace frames are sums until rt_pathtrace_resolve_device; "
                                             "convert the resolved frame with rt_quantize_device");
    return launch_render(c, (hipStream_t)stream, rgb_dev, hit_id_dev, false, rgb8_dev);
}

int rt_set_workgroups(rt_context* c, int32_t n_workgroups) {
    if (!c) return rt::set_error(RT_ERR_INVALID, "rt_set_workgroups: null context");
    if (n_workgroups < 0) return rt::set_error(RT_ERR_INVALID, "rt_set_workgroups: %d", n_workgroups);
    if (c->max_workgroups != n_workgroups) {
        c->max_workgroups = n_workgroups;
        ++c->generation;  // arena sizes were calibrated for the old grid
    }
    return RT_OK;
}

int rt_quantize_device(rt_context* c, void* stream, const double* rgb_dev, int64_t n_values, uint8_t* out_dev) {
    if (!c || !rgb_dev || !out_dev || n_values < 0) return rt::set_error(RT_ERR_INVALID, "rt_quantize_device: bad argument");
    if (n_values == 0) return RT_OK;
    HIP_TRY(hipSetDevice(c->device));
    long long grid = std::min<long long>((n_values + 255) / 256, (long long)c->n_cus * 16);
    hipLaunchKernelGGL(quantize_kernel, dim3((unsigned)grid), dim3(256), 0, (hipStream_t)stream, rgb_dev, (long long)n_values, out_dev);
    HIP_TRY(hipGetLastError());
    return RT_OK;
}

int rt_render_u8(rt_context* c, uint8_t* rgb8, int32_t* hit_id) {
    if (!c || !rgb8) return rt::set_error(RT_ERR_INVALID, "rt_render_u8: null argument");
    if (!c->have_params) return rt::set_error(RT_ERR_STATE, "rt_render_u8: params not set");
    HIP_TRY(hipSetDevice(c->device));
    const size_t n_px = (size_t)local_rows_of(c->params) * c->params.width;
    const bool pt = c->params.mode == RT_MODE_PATHTRACE;
    unsigned char* d_u8 = nullptr;
    double* d_rgb = nullptr;  // path trace: the sums, then the resolved frame
    int* d_hit = nullptr;
    auto release = [&]() {
        (void)hipFree(d_u8);
        (void)hipFree(d_rgb);
        (void)hipFree(d_hit);
    };
    hipError_t e = hipMalloc((void**)&d_u8, std::max<size_t>(n_px, 1) * 3);
    if (e == hipSuccess && pt) e = hipMalloc((void**)&d_rgb, std::max<size_t>(n_px, 1) * 3 * sizeof(double));
    if (e == hipSuccess && hit_id) {
        e = hipMalloc((void**)&d_hit, std::max<size_t>(n_px, 1) * sizeof(int));
        if (e == hipSuccess) e = hipMemset(d_hit, 0xFF, std::max<size_t>(n_px, 1) * sizeof(int));
    }
    if (e != hipSuccess) {
        release();
        return rt::set_error(RT_ERR_DEVICE, "rt_render_u8: hipMalloc -> %s", hipGetErrorString(e));
    }
    int rc = pt ? launch_render(c, nullptr, d_rgb, d_hit, false) : launch_render(c, nullptr, nullptr, d_hit, false, d_u8);
    if (rc == RT_OK && pt) rc = rt_quantize_device(c, nullptr, d_rgb, (int64_t)n_px * 3, d_u8);
    if (rc == RT_OK) {
        e = hipMemcpy(rgb8, d_u8, n_px * 3, hipMemcpyDeviceToHost);
        if (e == hipSuccess && hit_id) e = hipMemcpy(hit_id, d_hit, n_px * sizeof(int), hipMemcpyDeviceToHost);
        if (e != hipSuccess) rc = rt::set_error(RT_ERR_DEVICE, "rt_render_u8: kernel or copy failed -> %s", hipGetErrorString(e));
    }
    release();
    return rc;
}

int rt_render(rt_context* c, double* rgb, int32_t* hit_id) {
    if (!c || !rgb) return rt::set_error(RT_ERR_INVALID, "rt_render: null argument");
    if (!c->have_params) return rt::set_error(RT_ERR_STATE, "rt_render: params not set");
    HIP_TRY(hipSetDevice(c->device));
    size_t n_px = (size_t)local_rows_of(c->params) * c->params.width;
    double* d_rgb = nullptr;
    int* d_hit = nullptr;
    HIP_TRY(hipMalloc((void**)&d_rgb, std::max<size_t>(n_px, 1) * 3 * sizeof(double)));
    if (hit_id) {
        hipError_t e = hipMalloc((void**)&d_hit, std::max<size_t>(n_px, 1) * sizeof(int));
        if (e != hipSuccess) {
            (void)hipFree(d_rgb);
            return rt::set_error(RT_ERR_DEVICE, "rt_render: hipMalloc(hit_id) -> %s", hipGetErrorString(e));
        }
        (void)hipMemset(d_hit, 0xFF, std::max<size_t>(n_px, 1) * sizeof(int));
    }
    int rc = launch_render(c, nullptr, d_rgb, d_hit, false);
    if (rc == RT_OK) {
        hipError_t e = hipMemcpy(rgb, d_rgb, n_px * 3 * sizeof(double), hipMemcpyDeviceToHost);
        if (e == hipSuccess && hit_id) e = hipMemcpy(hit_id, d_hit, n_px * sizeof(int), hipMemcpyDeviceToHost);
        if (e != hipSuccess) rc = rt::set_error(RT_ERR_DEVICE, "rt_render: kernel or copy failed -> %s", hipGetErrorString(e));
    }
    (void)hipFree(d_rgb);
    (void)hipFree(d_hit);
    return rc;
}

int rt_get_stats(rt_context* c, rt_stats* out) {
    if (!c || !out) return rt::set_error(RT_ERR_INVALID, "rt_get_stats: null argument");
    std::memset(out, 0, sizeof *out);
    HIP_TRY(hipSetDevice(c->device));
    if (c->timed) {
        HIP_TRY(hipEventSynchronize(c->ev_stop));
        float ms = 0.f;
        HIP_TRY(hipEventElapsedTime(&ms, c->ev_start, c->ev_stop));
        out->kernel_ms = ms;
        std::vector<unsigned long long> rows((size_t)kCounterSlots * kCounterRow);
        HIP_TRY(hipMemcpy(rows.data(), c->d_counters, rows.size() * sizeof(unsigned long long), hipMemcpyDeviceToHost));
        unsigned long long cnt[8] = {0, 0, 0, 0, 0, 0, 0, 0};
        for (size_t r = 0; r < (size_t)kCounterSlots; ++r)
            for (int k = 0; k < 5; ++k) cnt[k] += rows[r * kCounterRow + k];
        out->rays = cnt[0];
        out->box_tests = cnt[1];
        out->sphere_tests = cnt[2];
        out->triangle_tests = cnt[3];
        out->cylinder_tests = cnt[4];
        out->prim_tests = cnt[2] + cnt[3] + cnt[4];
        if (c->last_was_wavefront && c->d_wf_state && c->params.max_depth > 0) {
            unsigned overflow = 0;
            if (c->level_kernels) {
                WfState st;
                HIP_TRY(hipMemcpy(&st, c->d_wf_state, sizeof st, hipMemcpyDeviceToHost));
                overflow = st.overflow;
            } else {
                HIP_TRY(hipMemcpy(&overflow, c->d_wf_state, sizeof overflow, hipMemcpyDeviceToHost));
            }
            if (overflow) {
                c->wf_calibrated_gen = 0;  // recalibrate on the next render
                return rt::set_error(RT_ERR_LIMIT, "the last render overflowed its node store; its output is invalid, render again");
            }
        }
    }
    out->upload_ms = c->upload_ms;
    out->n_bvh_nodes = c->n_bvh_nodes;
    out->scene_in_lds = c->scene_in_lds;
    return RT_OK;
}

// ---- diagnostics (declared in the "diagnostics" section of the header) --------------------------
int rt_debug_trace_rays(rt_context* c, int64_t n, const double* origins, const double* dirs, double* t_out,
                        int32_t* prim_out, double* normal_out) {
    if (!c || n < 0 || !origins || !dirs || !t_out || !prim_out || !normal_out)
        return rt::set_error(RT_ERR_INVALID, "rt_debug_trace_rays: bad argument");
    if (!c->have_scene) return rt::set_error(RT_ERR_STATE, "rt_debug_trace_rays: no scene");
    if (n == 0) return RT_OK;
    HIP_TRY(hipSetDevice(c->device));
    double *d_o = nullptr, *d_d = nullptr, *d_t = nullptr, *d_n = nullptr;
    int* d_p = nullptr;
    size_t v3 = (size_t)n * 3 * sizeof(double);
    int rc = RT_OK;
    hipError_t e = hipMalloc((void**)&d_o, v3);
    if (e == hipSuccess) e = hipMalloc((void**)&d_d, v3);
    if (e == hipSuccess) e = hipMalloc((void**)&d_t, (size_t)n * sizeof(double));
    if (e == hipSuccess) e = hipMalloc((void**)&d_n, v3);
    if (e == hipSuccess) e = hipMalloc((void**)&d_p, (size_t)n * sizeof(int));
    if (e == hipSuccess) e = hipMemcpy(d_o, origins, v3, hipMemcpyHostToDevice);
    if (e == hipSuccess) e = hipMemcpy(d_d, dirs, v3, hipMemcpyHostToDevice);
    if (e == hipSuccess) {
        int block = 256;
        long long grid = (n + block - 1) / block;
        hipLaunchKernelGGL(debug_trace_kernel, dim3((unsigned)grid), dim3(block), 0, 0, c->sv, c->traversal, (long long)n, d_o, d_d,
                           d_t, d_p, d_n);
        e = hipGetLastError();
    }
    if (e == hipSuccess) e = hipMemcpy(t_out, d_t, (size_t)n * sizeof(double), hipMemcpyDeviceToHost);
    if (e == hipSuccess) e = hipMemcpy(prim_out, d_p, (size_t)n * sizeof(int), hipMemcpyDeviceToHost);
    if (e == hipSuccess) e = hipMemcpy(normal_out, d_n, v3, hipMemcpyDeviceToHost);
    if (e != hipSuccess) rc = rt::set_error(RT_ERR_DEVICE, "rt_debug_trace_rays -> %s", hipGetErrorString(e));
    (void)hipFree(d_o); (void)hipFree(d_d); (void)hipFree(d_t); (void)hipFree(d_n); (void)hipFree(d_p);
    return rc;
}

int rt_debug_math(rt_context* c, int op, int64_t n, const double* a, const double* b, double* out) {
    if (!c || n < 0 || !a || !out) return rt::set_error(RT_ERR_INVALID, "rt_debug_math: bad argument");
    if (n == 0) return RT_OK;
    HIP_TRY(hipSetDevice(c->device));
    double *d_a = nullptr, *d_b = nullptr, *d_o = nullptr;
    size_t bytes = (size_t)n * sizeof(double);
    int rc = RT_OK;
    hipError_t e = hipMalloc((void**)&d_a, bytes);
    if (e == hipSuccess) e = hipMalloc((void**)&d_o, bytes);
    if (e == hipSuccess && b) e = hipMalloc((void**)&d_b, bytes);
    if (e == hipSuccess) e = hipMemcpy(d_a, a, bytes, hipMemcpyHostToDevice);
    if (e == hipSuccess && b) e = hipMemcpy(d_b, b, bytes, hipMemcpyHostToDevice);
    if (e == hipSuccess) {
        int block = 256;
        long long grid = (n + block - 1) / block;
        hipLaunchKernelGGL(debug_math_kernel, dim3((unsigned)grid), dim3(block), 0, 0, op, (long long)n, d_a, d_b, d_o);
        e = hipGetLastError();
    }
    if (e == hipSuccess) e = hipMemcpy(out, d_o, bytes, hipMemcpyDeviceToHost);
    if (e != hipSuccess) rc = rt::set_error(RT_ERR_DEVICE, "rt_debug_math -> %s", hipGetErrorString(e));
    (void)hipFree(d_a); (void)hipFree(d_b); (void)hipFree(d_o);
    return rc;
}

}  // extern "C"
