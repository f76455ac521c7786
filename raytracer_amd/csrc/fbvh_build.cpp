// fbvh_build.cpp -- builder of the device-side acceleration structure ("filter BVH").
//
// The reference's BVH (src/BVHNode.cpp, a median split that drags the two 2000-unit floor
// triangles of clutter_500 through nine levels of giant boxes) fixes WHAT a ray may hit and how ties
// resolve, but it does not have to be what the GPU walks.  For a regular ray (device_math.hpp:
// ray_regular) slab quotients are monotone in the box coordinates and a BVHNode's box is the exact
// union of its children's, so a primitive is reached by BVHNode::intersect if and only if the box
// of the node that holds it directly (its "gate": the span-1 leaf, or the span-2 node whose
// children are primitives, BVHNode.cpp:17-24) is hit.  Hence
//
//     reference result = among primitives whose gate box test AND primitive test pass,
//                        the smallest t; ties -> the last one in the reference's DFS leaf order
//                        (BVHNode.cpp:64-69); without a BVH: no gate, ties -> the first object
//                        (Scene.cpp:73).
//
// Everything else is free.  So the device walks THIS tree instead: a binned-SAH binary BVH over the
// primitives' TRUE bounds (conservatively enlarged), boxes in fp32, children ordered near-to-far per
// ray octant through precomputed hit/miss links (stackless), nodes culled by entry distance; the two
// exact fp64 tests of the reference (gate box, primitive) run only in its leaves.  The filter never
// decides a hit, it only decides where not to look, and it is conservative by construction.
#include <algorithm>
#include <cmath>
#include <cstring>
#include <limits>
#include <utility>
#include <vector>

#include "rt_internal.hpp"

namespace rt {

namespace {

struct Box3 {
    double lo[3], hi[3];
    void reset() {
        for (int k = 0; k < 3; ++k) {
            lo[k] = std::numeric_limits<double>::infinity();
            hi[k] = -std::numeric_limits<double>::infinity();
        }
    }
    void grow(const Box3& b) {
        for (int k = 0; k < 3; ++k) {
            lo[k] = std::min(lo[k], b.lo[k]);
            hi[k] = std::max(hi[k], b.hi[k]);
        }
    }
    double area() const {
        double e[3] = {std::max(hi[0] - lo[0], 0.0), std::max(hi[1] - lo[1], 0.0), std::max(hi[2] - lo[2], 0.0)};
        return 2.0 * (e[0] * e[1] + e[1] * e[2] + e[2] * e[0]);
    }
};

float round_down(double x) {
    float f = (float)x;
    if ((double)f > x) f = std::nextafterf(f, -std::numeric_limits<float>::infinity());
    return f;
}
float round_up(double x) {
    float f = (float)x;
    if ((double)f < x) f = std::nextafterf(f, std::numeric_limits<float>::infinity());
    return f;
}

struct BuildNode {
    Box3 box;
    int left = -1, right = -1, axis = 0;
    int first = 0, count = 0;  // leaf range in the ordered primitive list
};

struct SahBuilder {
    std::vector<Box3> pbox;
    std::vector<double> pcen;  // [n][3]
    std::vector<int> order;
    std::vector<BuildNode> nodes;
    static constexpr int kBins = 16;
    static constexpr int kMaxLeaf = 2;

    int build(int first, int count) {
        int idx = (int)nodes.size();
        nodes.emplace_back();
        Box3 bb, cb;
        bb.reset();
        cb.reset();
        for (int i = first; i < first + count; ++i) {
            int p = order[i];
            bb.grow(pbox[p]);
            for (int k = 0; k < 3; ++k) {
                cb.lo[k] = std::min(cb.lo[k], pcen[(size_t)p * 3 + k]);
                cb.hi[k] = std::max(cb.hi[k], pcen[(size_t)p * 3 + k]);
            }
        }
        nodes[idx].box = bb;
        nodes[idx].first = first;
        nodes[idx].count = count;
        if (count <= 1) return idx;
        // binned SAH over the three axes
        double best_cost = std::numeric_limits<double>::infinity();
        int best_axis = -1, best_split = -1;
        for (int ax = 0; ax < 3; ++ax) {
            double ext = cb.hi[ax] - cb.lo[ax];
            if (!(ext > 0)) continue;
            Box3 bin_box[kBins];
            int bin_cnt[kBins];
            for (int b = 0; b < kBins; ++b) {
                bin_box[b].reset();
                bin_cnt[b] = 0;
            }
            for (int i = first; i < first + count; ++i) {
                int p = order[i];
                int b = (int)((pcen[(size_t)p * 3 + ax] - cb.lo[ax]) / ext * kBins);
                b = std::min(std::max(b, 0), kBins - 1);
                bin_box[b].grow(pbox[p]);
                ++bin_cnt[b];
            }
            double right_area[kBins];
            int right_cnt[kBins];
            Box3 acc;
            acc.reset();
            int c = 0;
            for (int b = kBins - 1; b > 0; --b) {
                acc.grow(bin_box[b]);
                c += bin_cnt[b];
                right_area[b] = c ? acc.area() : 0.0;
                right_cnt[b] = c;
            }
            acc.reset();
            c = 0;
            for (int b = 0; b < kBins - 1; ++b) {
                acc.grow(bin_box[b]);
                c += bin_cnt[b];
                if (c == 0 || right_cnt[b + 1] == 0) continue;
                double cost = acc.area() * c + right_area[b + 1] * right_cnt[b + 1];
                if (cost < best_cost) {
                    best_cost = cost;
                    best_axis = ax;
                    best_split = b;
                }
            }
        }
        double leaf_cost = bb.area() * count;
        if (best_axis < 0 || (count <= kMaxLeaf && !(best_cost < leaf_cost))) {
            if (count <= kMaxLeaf) return idx;
            // all centroids coincide (or no useful split): halve the list
            best_axis = 0;
            int mid = first + count / 2;
            nodes[idx].axis = 0;
            nodes[idx].count = 0;
            int l = build(first, mid - first);
            int r = build(mid, first + count - mid);
            nodes[idx].left = l;
            nodes[idx].right = r;
            return idx;
        }
        double ext = cb.hi[best_axis] - cb.lo[best_axis];
        auto in_left = [&](int p) {
            int b = (int)((pcen[(size_t)p * 3 + best_axis] - cb.lo[best_axis]) / ext * kBins);
            b = std::min(std::max(b, 0), kBins - 1);
            return b <= best_split;
        };
        int mid = (int)(std::partition(order.begin() + first, order.begin() + first + count, in_left) - order.begin());
        if (mid == first || mid == first + count) mid = first + count / 2;
        nodes[idx].axis = best_axis;
        nodes[idx].count = 0;
        int l = build(first, mid - first);
        int r = build(mid, first + count - mid);
        nodes[idx].left = l;
        nodes[idx].right = r;
        return idx;
    }
};

void make_links(const std::vector<BuildNode>& nodes, int oct, int node, int next, std::vector<int32_t>& links) {
    const BuildNode& n = nodes[node];
    int32_t* l = &links[((size_t)node * 8 + oct) * 2];
    l[1] = next;  // miss
    if (n.left < 0) {
        l[0] = next;  // a leaf continues where a miss would
        return;
    }
    bool negative = (oct >> n.axis) & 1;  // direction component along the split axis is negative
    int near = negative ? n.right : n.left;
    int far = negative ? n.left : n.right;
    l[0] = near;
    make_links(nodes, oct, near, far, links);
    make_links(nodes, oct, far, next, links);
}

}  // namespace

void build_filter_bvh(const rt_scene_desc* s, const FlatBVH* ref_bvh, FilterBVH& out) {
    out = FilterBVH();
    const int n = s->n_prims;
    if (n <= 0) return;
    SahBuilder b;
    b.pbox.resize(n);
    b.pcen.resize((size_t)n * 3);
    b.order.resize(n);
    double scene_scale = 0.0;
    for (int i = 0; i < n; ++i) {
        const double* g = s->prim_geom + (size_t)i * RT_GEOM_STRIDE;
        Box3& bx = b.pbox[i];
        if (s->prim_type[i] == RT_SPHERE) {
            double r = std::fabs(g[3]);
            for (int k = 0; k < 3; ++k) {
                bx.lo[k] = g[k] - r;
                bx.hi[k] = g[k] + r;
            }
        } else if (s->prim_type[i] == RT_TRIANGLE) {
            for (int k = 0; k < 3; ++k) {
                bx.lo[k] = std::min({g[k], g[3 + k], g[6 + k]});
                bx.hi[k] = std::max({g[k], g[3 + k], g[6 + k]});
            }
        } else {
            // true extent of the capped cylinder the reference intersects (base .. base + axis * h, radius r;
            // src/Cylinder.cpp:13-107 -- not the +Y box of its getBoundingBox): along coordinate k the two cap
            // discs reach |a_k| h / 2 + r sqrt(1 - a_k^2) from the mid-point.  A bounding-sphere box, as used before,
            // is 5.6 times the volume for pathtrace.json's upright cylinder, and the cylinder test is the most
            // expensive one there is (150 flops, 4 divisions, a square root).
            double len = std::sqrt(g[3] * g[3] + g[4] * g[4] + g[5] * g[5]);
            double ax[3] = {g[3] / len, g[4] / len, g[5] / len};
            double h = g[7], r = std::fabs(g[6]);
            for (int k = 0; k < 3; ++k) {
                double c = g[k] + ax[k] * h * 0.5;
                double e = (std::fabs(ax[k]) * std::fabs(h) * 0.5 + r * std::sqrt(std::max(0.0, 1.0 - ax[k] * ax[k]))) * (1.0 + 1e-9) + 1e-300;
                bx.lo[k] = c - e;
                bx.hi[k] = c + e;
            }
        }
        for (int k = 0; k < 3; ++k) {
            b.pcen[(size_t)i * 3 + k] = 0.5 * (bx.lo[k] + bx.hi[k]);
            scene_scale = std::max({scene_scale, std::fabs(bx.lo[k]), std::fabs(bx.hi[k])});
        }
        b.order[i] = i;
    }
    bool finite = std::isfinite(scene_scale) && scene_scale > 0.0 && scene_scale <= 0x1p30;
    for (int i = 0; i < n && finite; ++i)
        for (int k = 0; k < 3; ++k) finite = finite && std::isfinite(b.pbox[i].lo[k]) && std::isfinite(b.pbox[i].hi[k]);
    if (!finite) return;  // out.ok stays false: the device uses the reference walk
    // Conservative enlargement of every primitive box (see DESIGN.md "Filter BVH: why it is safe"):
    //   1e-5 of the primitive's extent        -- computed hit points lie off the surface by rounding
    //   2^10 eps S^2 / extent                 -- quadratic/determinant cancellation for origins up to
    //                                            4S away (rays from farther out take the reference walk)
    //   2^-20 S                               -- the ray origin is rounded to fp32 on the device
    const double S = scene_scale;
    for (int i = 0; i < n; ++i) {
        Box3& bx = b.pbox[i];
        double ext = std::max({bx.hi[0] - bx.lo[0], bx.hi[1] - bx.lo[1], bx.hi[2] - bx.lo[2], 1e-300});
        double m = 1e-5 * ext + 1024.0 * 2.220446049250313e-16 * S * S / ext + 0x1p-20 * S;
        if (!(m < S)) m = S;
        for (int k = 0; k < 3; ++k) {
            bx.lo[k] -= m;
            bx.hi[k] += m;
        }
    }
    b.nodes.reserve((size_t)2 * n);
    b.build(0, n);
    const int nn = (int)b.nodes.size();
    {
        // Node order: the root first, then always the pending node with the largest surface area (the one a
        // random ray is likeliest to visit next), children after their parent.  A scene whose image does not
        // fit in LDS has the first part of this array staged there (render_hip.hip: StagedImg), so the prefix
        // should be the most visited treelet; for scenes that fit, the order does not matter.
        std::vector<std::pair<double, int>> heap;  // (area, old index), max-heap
        std::vector<int> order_of;                 // new index -> old index
        order_of.reserve(nn);
        heap.emplace_back(b.nodes[0].box.area(), 0);
        while (!heap.empty()) {
            std::pop_heap(heap.begin(), heap.end());
            const int old = heap.back().second;
            heap.pop_back();
            order_of.push_back(old);
            const BuildNode& bn = b.nodes[old];
            if (bn.left >= 0) {
                heap.emplace_back(b.nodes[bn.left].box.area(), bn.left);
                std::push_heap(heap.begin(), heap.end());
                heap.emplace_back(b.nodes[bn.right].box.area(), bn.right);
                std::push_heap(heap.begin(), heap.end());
            }
        }
        std::vector<int> new_of(nn, -1);
        for (int i = 0; i < nn; ++i) new_of[order_of[i]] = i;
        std::vector<BuildNode> re(nn);
        for (int i = 0; i < nn; ++i) {
            re[i] = b.nodes[order_of[i]];
            if (re[i].left >= 0) {
                re[i].left = new_of[re[i].left];
                re[i].right = new_of[re[i].right];
            }
        }
        b.nodes.swap(re);
    }
    out.n_nodes = nn;
    out.scene_scale = S;
    out.box.resize((size_t)nn * 6);
    out.leaf_first.resize(nn);
    out.leaf_count.resize(nn);
    out.links.assign((size_t)nn * 16, -1);
    for (int i = 0; i < nn; ++i) {
        const BuildNode& bn = b.nodes[i];
        for (int k = 0; k < 3; ++k) {
            out.box[(size_t)i * 6 + k] = round_down(bn.box.lo[k]);
            out.box[(size_t)i * 6 + 3 + k] = round_up(bn.box.hi[k]);
        }
        out.leaf_first[i] = bn.left < 0 ? bn.first : 0;
        out.leaf_count[i] = bn.left < 0 ? bn.count : 0;
    }
    for (int oct = 0; oct < 8; ++oct) make_links(b.nodes, oct, 0, -1, out.links);
    // leaf records: primitive ID, gate (reference node index) and tie-break key
    std::vector<int32_t> gate_of(n, -1), key_of(n, 0);
    if (ref_bvh && ref_bvh->size() > 0) {
        for (int i = 0; i < ref_bvh->size(); ++i) {
            if (ref_bvh->prim0[i] >= 0) {
                gate_of[ref_bvh->prim0[i]] = i;
                key_of[ref_bvh->prim0[i]] = 2 * i;  // pre-order index = DFS leaf order; later wins
            }
            if (ref_bvh->prim1[i] >= 0) {
                gate_of[ref_bvh->prim1[i]] = i;
                key_of[ref_bvh->prim1[i]] = 2 * i + 1;
            }
        }
    } else {
        for (int i = 0; i < n; ++i) key_of[i] = -i;  // linear scan: the first object wins
    }
    out.leaf_prim.resize((size_t)n * 4);
    out.leaf_box.resize((size_t)n * 6);
    for (int k = 0; k < n; ++k) {
        int p = b.order[k];
        for (int a = 0; a < 3; ++a) {
            out.leaf_box[(size_t)k * 6 + a] = round_down(b.pbox[p].lo[a]);
            out.leaf_box[(size_t)k * 6 + 3 + a] = round_up(b.pbox[p].hi[a]);
        }
        out.leaf_prim[(size_t)k * 4 + 0] = p;
        out.leaf_prim[(size_t)k * 4 + 1] = gate_of[p];
        out.leaf_prim[(size_t)k * 4 + 2] = key_of[p];
        out.leaf_prim[(size_t)k * 4 + 3] = 0;
    }
    out.ok = true;
}

}  // namespace rt
