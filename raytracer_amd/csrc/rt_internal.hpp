// rt_internal.hpp -- declarations shared by the host ingest (host_scene.cpp) and the device
// side (render_hip.hip) of libraytracer_amd.so.  Not part of the C-ABI.
#pragma once
#include <cstdarg>
#include <cstdint>
#include <string>
#include <vector>

#include "raytracer_amd.h"

namespace rt {

// thread-local message behind rt_last_error()
int set_error(int code, const char* fmt, ...) __attribute__((format(printf, 2, 3)));

// Flattened BVHNode tree (src/BVHNode.cpp:7-45) in DFS pre-order.  Because the reference visits
// both children of every node it enters, in a fixed left-then-right order, the traversal is a
// fixed walk over this array: at node i go to i+1 when the box is hit and to skip[i] when it is
// missed (for a leaf skip[i] == i+1).  prim0/prim1 are the primitives tested at the node:
//   span 1 leaf  -> prim0 = object, prim1 = -1
//   span 2 node  -> prim0 = left,   prim1 = right   (no boxes of their own, BVHNode.cpp:21-24)
//   span >= 3    -> prim0 = prim1 = -1
struct FlatBVH {
    std::vector<double> box;     // [n][6] min xyz, max xyz
    std::vector<int32_t> skip;   // [n]
    std::vector<int32_t> prim0;  // [n] primitive ID or -1
    std::vector<int32_t> prim1;  // [n]
    std::vector<int32_t> right;  // [n] index of the right child node (span >= 3) or -1
    int size() const { return (int)skip.size(); }
};

// Builds the tree exactly as the reference constructor does (same axis rule, same std::sort on
// the same comparison sequence, same span/2 split).
void build_flat_bvh(const rt_scene_desc* scene, FlatBVH& out);

// Device-side acceleration structure (fbvh_build.cpp): SAH binary BVH over true primitive bounds,
// fp32 boxes, per-octant hit/miss links, leaves carrying (primitive, gate, tie key).
struct FilterBVH {
    bool ok = false;
    int n_nodes = 0;
    double scene_scale = 0.0;          // max |coordinate| of the primitives' true bounds
    std::vector<float> box;            // [n][6] lo xyz, hi xyz (already enlarged, rounded outward)
    std::vector<int32_t> leaf_first;   // [n] first record in leaf_prim (leaves), 0 otherwise
    std::vector<int32_t> leaf_count;   // [n] number of records (0 for inner nodes)
    std::vector<int32_t> links;        // [n][8 octants][2]: next node when the box passes / fails; -1 = done
    std::vector<int32_t> leaf_prim;    // [n_prims][4]: primitive ID, gate = reference node index or -1, tie key, 0
    std::vector<float> leaf_box;       // [n_prims][6], in leaf_prim's order: the primitive's own enlarged box, lo xyz, hi xyz, rounded outward
};
void build_filter_bvh(const rt_scene_desc* scene, const FlatBVH* ref_bvh, FilterBVH& out);

// Per-primitive AABBs as the reference's getBoundingBox() returns them, including the
// cylinder box that assumes a +Y axis (src/Cylinder.cpp:131-137).
void prim_bounds(const rt_scene_desc* scene, int prim, double box[6]);

// Triangle::Triangle normal (src/Triangle.cpp:7-12).
void triangle_normal(const double* v0, const double* v1, const double* v2, double n[3]);

// Host-side handles of the frame kernels, one translation unit per family (k_blockwave.hip, k_fused.hip,
// k_pathwave.hip); every one has the signature (SceneView, RenderArgs, BwArgs) and is launched through
// hipLaunchKernel.  tv = traversal - 1 (0 walk, 1 cull, 2 filter BVH).
const void* bw_kernel_ptr(bool path_trace, bool lds, int tv);
const void* fw_kernel_ptr(bool path_trace, bool lds, int tv);
const void* fw_roots_kernel_ptr(bool lds, int tv);
const void* pt_kernel_ptr(bool lds, int tv);
const void* pq_kernel_ptr(int shadow_rays_per_hit, int waves);  // (SceneView, RenderArgs, PqArgs): path trace out of per-wave queues, LDS-resident scenes, filter-BVH traversal

}  // namespace rt
