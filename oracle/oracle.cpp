// oracle/oracle.cpp -- TEST INFRASTRUCTURE, NOT PRODUCT CODE.
//
// A from-scratch fp64 CPU restatement of the reference render loop (Basemism/Raytracer),
// written the way the reference is shaped (recursive shading, pointer-linked BVH, both children
// always visited) so that it shares no structure with the flattened / iterative HIP path it
// checks.  Only tests/, __graft_entry__.smoke() and bench.py's cpu_baseline leg may load this
// library; the product (raytracer_amd/, the CLI) never does.
//
// Pinning: tests/test_oracle_pinned.py checks this file's PPM output byte-for-byte against the
// sha256 values of the unmodified reference (oracle/_ref, built by `make -C oracle ref`) on
// binary_primitives, blinn_phong, mirror_glass_cylinder and clutter_500, and
// its ray counts against the reference's Scene::intersect call counts.  Path tracing cannot be
// pinned bit-wise (the reference seeds from random_device, RayTracer.cpp:34, AreaLight.cpp:4,
// Camera.h:39): it is pinned statistically against a committed high-spp render of oracle/_ref
// (tests/golden/make_golden.py).
//
// Every function cites the reference lines it restates (paths relative to /root/reference).
// The only deliberate departure is the random source: a counter-based Philox4x32-10 keyed by
// (seed; pixel, sample, tree-node, dimension), defined in DESIGN.md "RNG", replaces the shared
// racing mt19937 / rand() of the reference.  Build: -O2 -ffp-contract=off (no FMA, as the
// reference's x86-64 build has none).

#include <algorithm>
#include <atomic>
#include <cmath>
#include <cstdint>
#include <cstring>
#include <limits>
#include <memory>
#include <vector>
#ifdef _OPENMP
#include <omp.h>
#endif

#include "raytracer_amd.h"

namespace {

// ---------------------------------------------------------------------------------------------
// Vector3 (src/Vector3.cpp:16-81).  Operation order is kept literally: dot is x*x'+y*y'+z*z'
// left to right, normalize is three divides by length, length is sqrt of the double sum.
struct V3 {
    double x, y, z;
    V3() : x(0), y(0), z(0) {}
    V3(double a, double b, double c) : x(a), y(b), z(c) {}
    explicit V3(const double* p) : x(p[0]), y(p[1]), z(p[2]) {}
    double operator[](int i) const { return i == 0 ? x : (i == 1 ? y : z); }
};
inline V3 operator+(const V3& a, const V3& b) { return V3(a.x + b.x, a.y + b.y, a.z + b.z); }
inline V3 operator-(const V3& a, const V3& b) { return V3(a.x - b.x, a.y - b.y, a.z - b.z); }
inline V3 operator*(const V3& a, const V3& b) { return V3(a.x * b.x, a.y * b.y, a.z * b.z); }
inline V3 operator/(const V3& a, const V3& b) { return V3(a.x / b.x, a.y / b.y, a.z / b.z); }
inline V3 operator*(const V3& a, double s) { return V3(a.x * s, a.y * s, a.z * s); }
inline V3 operator/(const V3& a, double s) { return V3(a.x / s, a.y / s, a.z / s); }
inline V3 operator+(const V3& a, double s) { return V3(a.x + s, a.y + s, a.z + s); }
inline V3 operator-(const V3& a, double s) { return V3(a.x - s, a.y - s, a.z - s); }
inline V3 operator-(const V3& a) { return V3(-a.x, -a.y, -a.z); }
inline double dot(const V3& a, const V3& b) { return a.x * b.x + a.y * b.y + a.z * b.z; }
inline V3 cross(const V3& a, const V3& b) {
    return V3(a.y * b.z - a.z * b.y, a.z * b.x - a.x * b.z, a.x * b.y - a.y * b.x);
}
inline double length(const V3& a) { return std::sqrt(a.x * a.x + a.y * a.y + a.z * a.z); }
inline V3 normalize(const V3& a) {
    double len = length(a);
    return V3(a.x / len, a.y / len, a.z / len);
}

// Ray (src/Ray.cpp:5-11): the constructor normalizes.
struct Ray {
    V3 o, d;
    Ray(const V3& origin, const V3& dir) : o(origin), d(normalize(dir)) {}
    V3 at(double t) const { return o + d * t; }
};

// ---------------------------------------------------------------------------------------------
// Philox4x32-10 (Salmon et al., SC'11).  Known-answer vectors are checked in
// tests/test_oracle_units.py through oracle_philox().
inline void philox4x32_10(const uint32_t ctr[4], const uint32_t key[2], uint32_t out[4]) {
    uint32_t c0 = ctr[0], c1 = ctr[1], c2 = ctr[2], c3 = ctr[3];
    uint32_t k0 = key[0], k1 = key[1];
    for (int r = 0; r < 10; ++r) {
        uint64_t p0 = (uint64_t)0xD2511F53u * c0;
        uint64_t p1 = (uint64_t)0xCD9E8D57u * c2;
        uint32_t n0 = (uint32_t)(p1 >> 32) ^ c1 ^ k0;
        uint32_t n1 = (uint32_t)p1;
        uint32_t n2 = (uint32_t)(p0 >> 32) ^ c3 ^ k1;
        uint32_t n3 = (uint32_t)p0;
        c0 = n0; c1 = n1; c2 = n2; c3 = n3;
        k0 += 0x9E3779B9u;
        k1 += 0xBB67AE85u;
    }
    out[0] = c0; out[1] = c1; out[2] = c2; out[3] = c3;
}

// Uniform double in [0,1) for (pixel, sample, node, dim): block dim/2, half dim%2, 53 bits.
struct Rng {
    uint32_t key[2];
    uint32_t pixel, sample;
    double u(uint32_t node, uint32_t dim) const {
        uint32_t ctr[4] = {pixel, sample, node, dim >> 1};
        uint32_t r[4];
        philox4x32_10(ctr, key, r);
        uint64_t bits = (dim & 1) ? (((uint64_t)r[3] << 32) | r[2]) : (((uint64_t)r[1] << 32) | r[0]);
        return (double)(bits >> 11) * 0x1.0p-53;
    }
};
// Dimension map (DESIGN.md "RNG").  Camera node = 0; the hit reached by the primary ray is
// node 1; the reflect/mirror/diffuse child of node c is 2c, the refract child 2c+1.
enum : uint32_t {
    DIM_JITTER_X = 0, DIM_JITTER_Y = 1, DIM_LENS0 = 2,  // camera node
    DIM_RR = 0, DIM_HEMI_1 = 2, DIM_HEMI_2 = 3, DIM_LIGHT0 = 4  // hit nodes (the two hemisphere dimensions share one counter block)
};

// ---------------------------------------------------------------------------------------------
struct Box {  // include/BoundingBox.h
    V3 mn, mx;
};
inline Box merge(const Box& a, const Box& b) {  // src/BoundingBox.cpp:10-22
    return Box{V3(std::min(a.mn.x, b.mn.x), std::min(a.mn.y, b.mn.y), std::min(a.mn.z, b.mn.z)),
               V3(std::max(a.mx.x, b.mx.x), std::max(a.mx.y, b.mx.y), std::max(a.mx.z, b.mx.z))};
}
inline V3 center(const Box& b) { return (b.mn + b.mx) * 0.5; }  // src/BoundingBox.cpp:64-66

// src/BoundingBox.cpp:24-62.  True divisions; a 0/0 NaN makes every compare false.
inline bool box_hit(const Box& b, const Ray& r) {
    double tmin = (b.mn.x - r.o.x) / r.d.x;
    double tmax = (b.mx.x - r.o.x) / r.d.x;
    if (tmin > tmax) std::swap(tmin, tmax);
    double tymin = (b.mn.y - r.o.y) / r.d.y;
    double tymax = (b.mx.y - r.o.y) / r.d.y;
    if (tymin > tymax) std::swap(tymin, tymax);
    if ((tmin > tymax) || (tymin > tmax)) return false;
    if (tymin > tmin) tmin = tymin;
    if (tymax < tmax) tmax = tymax;
    double tzmin = (b.mn.z - r.o.z) / r.d.z;
    double tzmax = (b.mx.z - r.o.z) / r.d.z;
    if (tzmin > tzmax) std::swap(tzmin, tzmax);
    if ((tmin > tzmax) || (tzmin > tmax)) return false;
    if (tzmin > tmin) tmin = tzmin;
    if (tzmax < tmax) tmax = tzmax;
    return tmax > 0;
}

struct Hit {  // include/Intersectable.h:16-26 with the Material copy replaced by the prim ID
    double t = 0.0;
    V3 point, normal;
    int prim = -1;
};

struct Prim {
    int type;
    int material;
    // sphere: c, r.  triangle: v0 v1 v2 n.  cylinder: base, axis, r, h.
    V3 a, b, c, n;
    double r = 0, h = 0;
};

struct Texture {
    int w = 0, h = 0;
    std::vector<V3> texel;  // byte/255.0, src/Material.cpp:63
};

struct Light {
    int type;
    V3 intensity, position, normal, u, v;
    double width, height;
};

struct Counters {
    uint64_t rays = 0, boxes = 0, prims = 0;
};

// Sphere (src/Sphere.cpp:14-63)
bool sphere_hit(const Prim& s, const Ray& ray, Hit& h) {
    V3 oc = ray.o - s.a;
    double a = dot(ray.d, ray.d);
    double b = 2.0 * dot(oc, ray.d);
    double c = dot(oc, oc) - s.r * s.r;
    double disc = b * b - 4 * a * c;
    if (disc < 0) return false;
    double sq = std::sqrt(disc);
    double t0 = (-b - sq) / (2.0 * a);
    double t1 = (-b + sq) / (2.0 * a);
    double t = t0;
    if (t < 0) {
        t = t1;
        if (t < 0) return false;
    }
    h.t = t;
    h.point = ray.at(t);
    h.normal = normalize(h.point - s.a);
    return true;
}
void sphere_uv(const Prim& s, const V3& point, double& u, double& v) {
    V3 p = normalize(point - s.a);
    double phi = atan2(p.z, p.x);
    double theta = acos(p.y);
    u = (phi + M_PI) / (2 * M_PI);
    v = theta / M_PI;
}
Box sphere_box(const Prim& s) {
    V3 rv(s.r, s.r, s.r);
    return Box{s.a - rv, s.a + rv};
}

// Triangle (src/Triangle.cpp:5-94)
V3 triangle_normal(const V3& v0, const V3& v1, const V3& v2) {
    V3 n = normalize(cross(v1 - v0, v2 - v0));
    if (dot(n, v0) > 0) n = -n;  // flipped towards the world origin, not the camera
    return n;
}
bool triangle_hit(const Prim& tr, const Ray& ray, Hit& h) {
    const double EPS = 1e-8;
    V3 e1 = tr.b - tr.a;
    V3 e2 = tr.c - tr.a;
    V3 hv = cross(ray.d, e2);
    double a = dot(e1, hv);
    if (std::abs(a) < EPS) return false;
    double f = 1.0 / a;
    V3 s = ray.o - tr.a;
    double u = f * dot(s, hv);
    if (u < 0.0 || u > 1.0) return false;
    V3 q = cross(s, e1);
    double v = f * dot(ray.d, q);
    if (v < 0.0 || u + v > 1.0) return false;
    double t = f * dot(e2, q);
    if (t > EPS) {
        h.t = t;
        h.point = ray.at(t);
        h.normal = tr.n;
        return true;
    }
    return false;
}
void triangle_uv(const Prim& tr, const V3& point, double& u, double& v) {
    V3 e1 = tr.b - tr.a;
    V3 e2 = tr.c - tr.a;
    V3 pv = point - tr.a;
    double d00 = dot(e1, e1), d01 = dot(e1, e2), d11 = dot(e2, e2);
    double d20 = dot(pv, e1), d21 = dot(pv, e2);
    double denom = d00 * d11 - d01 * d01;
    double vc = (d11 * d20 - d01 * d21) / denom;
    double wc = (d00 * d21 - d01 * d20) / denom;
    double uc = 1.0 - vc - wc;
    u = (uc + vc) / 2.0;
    v = (vc + wc) / 2.0;
}
Box triangle_box(const Prim& t) {
    return Box{V3(std::min({t.a.x, t.b.x, t.c.x}), std::min({t.a.y, t.b.y, t.c.y}),
                  std::min({t.a.z, t.b.z, t.c.z})),
               V3(std::max({t.a.x, t.b.x, t.c.x}), std::max({t.a.y, t.b.y, t.c.y}),
                  std::max({t.a.z, t.b.z, t.c.z}))};
}

// Cylinder (src/Cylinder.cpp:13-137); a = baseCenter, b = axis.
bool cylinder_hit(const Prim& cy, const Ray& ray, Hit& h) {
    V3 oc = ray.o - cy.a;
    V3 axis = cy.b;
    V3 d = ray.d - axis * dot(ray.d, axis);
    V3 ocp = oc - axis * dot(oc, axis);
    double a = dot(d, d);
    double b = 2.0 * dot(d, ocp);
    double c = dot(ocp, ocp) - cy.r * cy.r;
    double disc = b * b - 4 * a * c;
    double t = INFINITY;
    V3 normal;
    bool hit = false;
    if (disc >= 0) {
        double sq = std::sqrt(disc);
        double t0 = (-b - sq) / (2 * a);
        double t1 = (-b + sq) / (2 * a);
        if (t0 > t1) std::swap(t0, t1);
        double y0 = dot(ray.o + ray.d * t0 - cy.a, axis);
        if (t0 >= 0 && y0 >= 0 && y0 <= cy.h) {
            t = t0;
            V3 p = ray.at(t);
            normal = normalize(p - cy.a - axis * y0);
            hit = true;
        } else {
            double y1 = dot(ray.o + ray.d * t1 - cy.a, axis);
            if (t1 >= 0 && y1 >= 0 && y1 <= cy.h) {
                t = t1;
                V3 p = ray.at(t);
                normal = normalize(p - cy.a - axis * y1);
                hit = true;
            }
        }
    }
    // caps (hasCaps is always true: the parse never passes the flag, RayTracer.cpp:991)
    double tb = dot(cy.a - ray.o, axis) / dot(ray.d, axis);
    if (tb >= 0) {
        V3 p = ray.at(tb);
        V3 dd = p - cy.a;
        if (dot(dd, dd) <= cy.r * cy.r) {
            if (tb < t) {
                t = tb;
                normal = -axis;
                hit = true;
            }
        }
    }
    V3 top = cy.a + axis * cy.h;
    double tt = dot(top - ray.o, axis) / dot(ray.d, axis);
    if (tt >= 0) {
        V3 p = ray.at(tt);
        V3 dd = p - top;
        if (dot(dd, dd) <= cy.r * cy.r) {
            if (tt < t) {
                t = tt;
                normal = axis;
                hit = true;
            }
        }
    }
    if (hit) {
        h.t = t;
        h.point = ray.at(t);
        h.normal = normal;
        return true;
    }
    return false;
}
void cylinder_uv(const Prim& cy, const V3& point, double& u, double& v) {
    V3 p = point - cy.a;
    double y = dot(p, cy.b);
    V3 pp = p - cy.b * y;
    double theta = atan2(pp.z, pp.x);
    if (theta < 0) theta += 2 * M_PI;
    u = theta / (2 * M_PI);
    v = y / cy.h;
    v = std::clamp(v, 0.0, 1.0);
}
Box cylinder_box(const Prim& cy) {  // assumes axis = +Y whatever the axis is (Cylinder.cpp:131-137)
    V3 rv(cy.r, cy.r, cy.r);
    V3 hv(0, cy.h, 0);
    return Box{cy.a - rv, cy.a + rv + hv};
}

// ---------------------------------------------------------------------------------------------
struct Node;  // BVHNode (include/BVHNode.h)
struct Child {
    Node* node = nullptr;  // BVHNode child (span >= 3 parents)
    int prim = -1;         // primitive child (span == 2 parents) or leaf object
};
struct Node {
    Box box;
    Child left, right;
    int object = -1;  // leaf (span 1)
    bool leaf = false;
};

struct Scene {
    std::vector<Prim> prims;
    std::vector<rt_material> mats;
    std::vector<Texture> texs;
    std::vector<Light> lights;
    V3 background;
    std::vector<std::unique_ptr<Node>> pool;
    Node* root = nullptr;

    Box prim_box(int i) const {
        const Prim& p = prims[i];
        return p.type == RT_SPHERE ? sphere_box(p) : (p.type == RT_TRIANGLE ? triangle_box(p) : cylinder_box(p));
    }
    bool prim_hit(int i, const Ray& r, Hit& h, Counters& c) const {
        const Prim& p = prims[i];
        ++c.prims;
        bool ok = p.type == RT_SPHERE ? sphere_hit(p, r, h)
                                      : (p.type == RT_TRIANGLE ? triangle_hit(p, r, h) : cylinder_hit(p, r, h));
        if (ok) h.prim = i;
        return ok;
    }
    void prim_uv(int i, const V3& pt, double& u, double& v) const {
        const Prim& p = prims[i];
        if (p.type == RT_SPHERE) sphere_uv(p, pt, u, v);
        else if (p.type == RT_TRIANGLE) triangle_uv(p, pt, u, v);
        else cylinder_uv(p, pt, u, v);
    }

    // BVHNode::BVHNode (src/BVHNode.cpp:7-45): `objs` is the (sub)sequence of primitive IDs.
    Node* build(const std::vector<int>& objs, size_t start, size_t end) {
        pool.emplace_back(new Node());
        Node* n = pool.back().get();
        Box bb = prim_box(objs[start]);
        for (size_t i = start + 1; i < end; ++i) bb = merge(bb, prim_box(objs[i]));
        n->box = bb;
        size_t span = end - start;
        if (span == 1) {
            n->leaf = true;
            n->object = objs[start];
        } else if (span == 2) {
            n->left.prim = objs[start];
            n->right.prim = objs[start + 1];
        } else {
            V3 size = bb.mx - bb.mn;
            int axis = 0;
            if (size.y > size.x) axis = 1;
            if (size.z > size[axis]) axis = 2;
            std::vector<int> sorted(objs.begin() + start, objs.begin() + end);
            std::sort(sorted.begin(), sorted.end(), [this, axis](int a, int b) {
                return center(prim_box(a))[axis] < center(prim_box(b))[axis];
            });
            size_t mid = span / 2;
            n->left.node = build(sorted, 0, mid);
            n->right.node = build(sorted, mid, sorted.size());
        }
        return n;
    }

    // BVHNode::intersect (src/BVHNode.cpp:47-81)
    bool node_hit(const Node* n, const Ray& r, Hit& out, Counters& c) const {
        ++c.boxes;
        if (!box_hit(n->box, r)) return false;
        if (n->leaf) return prim_hit(n->object, r, out, c);
        Hit lh, rh;
        bool hl = n->left.node ? node_hit(n->left.node, r, lh, c) : prim_hit(n->left.prim, r, lh, c);
        bool hr = n->right.node ? node_hit(n->right.node, r, rh, c) : prim_hit(n->right.prim, r, rh, c);
        if (hl && hr) {
            out = (lh.t < rh.t) ? lh : rh;
            return true;
        }
        if (hl) { out = lh; return true; }
        if (hr) { out = rh; return true; }
        return false;
    }

    // Scene::intersect (src/Scene.cpp:63-81)
    bool intersect(const Ray& r, Hit& out, Counters& c) const {
        ++c.rays;
        if (root) return node_hit(root, r, out, c);
        bool any = false;
        double closest = std::numeric_limits<double>::max();
        Hit tmp;
        for (int i = 0; i < (int)prims.size(); ++i) {
            if (prim_hit(i, r, tmp, c) && tmp.t < closest) {
                any = true;
                closest = tmp.t;
                out = tmp;
            }
        }
        return any;
    }

    // Material::getTextureColor (src/Material.cpp:69-86)
    V3 texture_color(const rt_material& m, double u, double v) const {
        if (m.texture < 0 || texs[m.texture].texel.empty()) return V3(m.diffuse);
        const Texture& t = texs[m.texture];
        u = u - floor(u);
        v = v - floor(v);
        int x = static_cast<int>(u * t.w);
        int y = static_cast<int>((1.0 - v) * t.h);
        x = std::clamp(x, 0, t.w - 1);
        y = std::clamp(y, 0, t.h - 1);
        return t.texel[(size_t)y * t.w + x];
    }
};

void load_scene(const rt_scene_desc* d, Scene& s) {
    s.prims.resize(d->n_prims);
    for (int i = 0; i < d->n_prims; ++i) {
        const double* g = d->prim_geom + (size_t)i * RT_GEOM_STRIDE;
        Prim& p = s.prims[i];
        p.type = d->prim_type[i];
        p.material = d->prim_material[i];
        if (p.type == RT_SPHERE) {
            p.a = V3(g);
            p.r = g[3];
        } else if (p.type == RT_TRIANGLE) {
            p.a = V3(g); p.b = V3(g + 3); p.c = V3(g + 6);
            p.n = triangle_normal(p.a, p.b, p.c);
        } else {
            p.a = V3(g);
            p.b = normalize(V3(g + 3));  // Cylinder ctor normalizes (again), src/Cylinder.cpp:11
            p.r = g[6];
            p.h = g[7];
        }
    }
    s.mats.assign(d->materials, d->materials + d->n_materials);
    s.texs.resize(d->n_textures);
    for (int i = 0; i < d->n_textures; ++i) {
        Texture& t = s.texs[i];
        t.w = d->textures[i].width;
        t.h = d->textures[i].height;
        size_t n = (size_t)t.w * t.h;
        t.texel.resize(n);
        for (size_t k = 0; k < n; ++k) {
            const uint8_t* px = d->textures[i].rgb + 3 * k;
            t.texel[k] = V3(px[0] / 255.0, px[1] / 255.0, px[2] / 255.0);
        }
    }
    s.lights.resize(d->n_lights);
    for (int i = 0; i < d->n_lights; ++i) {
        const rt_light& l = d->lights[i];
        Light& o = s.lights[i];
        o.type = l.type;
        o.intensity = V3(l.intensity);
        o.position = V3(l.position);
        if (l.type == RT_LIGHT_AREA) {  // include/AreaLight.h:21-22
            o.normal = normalize(V3(l.normal));
            o.u = normalize(V3(l.u));
            o.v = normalize(V3(l.v));
        }
        o.width = l.width;
        o.height = l.height;
    }
    s.background = V3(d->background);
    if (d->use_bvh && d->n_prims > 0) {
        std::vector<int> ids(d->n_prims);
        for (int i = 0; i < d->n_prims; ++i) ids[i] = i;
        s.root = s.build(ids, 0, ids.size());
    }
}

// ---------------------------------------------------------------------------------------------
struct Cam {
    V3 position, lower_left, horizontal, vertical, u, v;
    double lens_radius;
};

// Camera::getRay (src/Camera.cpp:33-49) + randomInUnitDisk (include/Camera.h:37-46)
Ray camera_ray(const Cam& c, double s, double t, bool use_lens, const Rng* rng) {
    V3 offset(0, 0, 0);
    if (use_lens && c.lens_radius > 0.0) {
        double x, y;
        uint32_t k = 0;
        do {
            x = -1.0 + 2.0 * rng->u(0, DIM_LENS0 + 2 * k);
            y = -1.0 + 2.0 * rng->u(0, DIM_LENS0 + 2 * k + 1);
            ++k;
        } while (x * x + y * y >= 1.0);
        V3 rd = V3(x, y, 0) * c.lens_radius;
        offset = c.u * rd.x + c.v * rd.y;
    }
    V3 image_point = c.lower_left + c.horizontal * s + c.vertical * t;
    V3 dir = image_point - c.position - offset;
    return Ray(c.position + offset, normalize(dir));
}

// toneMap (RayTracer.cpp:161-202).  Float literals widened to double, as written there.
V3 uncharted2_partial(V3 x) {
    float A = 0.15f, B = 0.50f, C = 0.10f, D = 0.20f, E = 0.02f, F = 0.30f;
    return ((x * (x * A + C * B) + D * E) / (x * (x * A + B) + D * F)) - E / F;
}
V3 tone_map(V3 color, int tm) {
    if (tm == RT_TONEMAP_REINHARD) return color / (color + V3(1, 1, 1));
    if (tm == RT_TONEMAP_WARD) {
        color = color / (color + V3(1, 1, 1));
        return color * (color * (color * 0.4 + 0.4) + 1.219) / (color * (color * 0.3 + 0.5) + 2.219);
    }
    if (tm == RT_TONEMAP_UNCHARTED2) {
        float exposure_bias = 2.0f;
        V3 curr = uncharted2_partial(color * exposure_bias);
        V3 W(11.2f, 11.2f, 11.2f);
        V3 white_scale = V3(1.0f, 1.0f, 1.0f) / uncharted2_partial(W);
        return curr * white_scale;
    }
    return color;
}

struct Renderer {
    const Scene& sc;
    const rt_params& P;
    Renderer(const Scene& s, const rt_params& p) : sc(s), P(p) {}

    // ---- Whitted ---------------------------------------------------------------------------
    // fresnelReflectance (RayTracer.cpp:500-504)
    static double schlick(double cosTheta, double ior) {
        double r0 = (1.0 - ior) / (1.0 + ior);
        r0 = r0 * r0;
        return r0 + (1.0 - r0) * pow(1.0 - cosTheta, 5.0);
    }

    // traceRay (RayTracer.cpp:373-393); first_prim receives the hit ID at depth 0.
    V3 trace(const Ray& ray, int depth, Counters& c, int* first_prim) const {
        if (depth >= P.max_depth) return sc.background;
        Hit h;
        if (sc.intersect(ray, h, c)) {
            if (first_prim) *first_prim = h.prim;
            if (P.mode == RT_MODE_PHONG) return shade_phong(h, ray, depth, c);
            if (P.mode == RT_MODE_BINARY) return V3(1.0, 0.0, 0.0);
            return sc.background;
        }
        if (first_prim) *first_prim = -1;
        if (P.mode == RT_MODE_BINARY) return V3(0, 0, 0);
        return sc.background;
    }

    // computeShadingPhong (RayTracer.cpp:694-804)
    V3 shade_phong(const Hit& h, const Ray& ray, int depth, Counters& c) const {
        const rt_material& m = sc.mats[sc.prims[h.prim].material];
        V3 tex(m.diffuse);
        if (m.texture >= 0) {
            double u, v;
            sc.prim_uv(h.prim, h.point, u, v);
            tex = sc.texture_color(m, u, v);
        }
        V3 ambient = tex * 0.25;
        V3 diffuse(0, 0, 0), specular(0, 0, 0);
        V3 view = -normalize(ray.d);
        for (const Light& L : sc.lights) {
            // Light::getPosition is only overridden by PointLight (src/Light.cpp:9-11)
            V3 lpos = L.type == RT_LIGHT_POINT ? L.position : V3(0, 0, 0);
            V3 ldir = normalize(lpos - h.point);
            V3 half = normalize(ldir + view);
            Ray shadow(h.point + h.normal * P.shadow_bias, ldir);
            Hit sh;
            bool in_shadow = false;
            if (sc.intersect(shadow, sh, c)) {
                double ldist = length(lpos - h.point);
                if (sh.t < ldist) in_shadow = true;
            }
            if (!in_shadow) {
                double df = std::max(0.0, dot(h.normal, ldir));
                diffuse = diffuse + tex * m.kd * df * L.intensity;
                double sf = pow(std::max(0.0, dot(h.normal, half)), (double)m.specular_exponent);
                specular = specular + V3(m.specular) * m.ks * sf * L.intensity;
            }
        }
        V3 local = ambient + diffuse + specular;
        if (m.is_reflective) {
            V3 n = h.normal;
            if (dot(ray.d, n) > 0.0) n = -n;
            V3 rdir = ray.d - n * 2 * dot(ray.d, n);
            Ray rr(h.point + n * P.shadow_bias, rdir);
            V3 rc = trace(rr, depth + 1, c, nullptr);
            local = local * (1 - m.reflectivity) + rc * m.reflectivity;
        }
        if (m.is_refractive && depth < P.max_depth) {
            double n1 = 1.0, n2 = m.refractive_index;
            V3 n = h.normal;
            if (dot(ray.d, n) > 0.0) {
                n = -n;
                std::swap(n1, n2);
            }
            double eta = n1 / n2;
            double cosI = -dot(n, ray.d);
            double sinT2 = eta * eta * (1.0 - cosI * cosI);
            if (sinT2 <= 1.0) {
                double cosT = std::sqrt(1.0 - sinT2);
                V3 tdir = normalize(ray.d * eta + n * (eta * cosI - cosT));
                double R = schlick(cosI, n2);
                Ray tr(h.point - n * P.shadow_bias, tdir);
                V3 tc = trace(tr, depth + 1, c, nullptr);
                V3 rdir = ray.d - n * 2.0 * dot(ray.d, n);
                Ray rr(h.point + n * P.shadow_bias, rdir);
                V3 rc = trace(rr, depth + 1, c, nullptr);
                local = rc * R + tc * (1.0 - R);
            }
        }
        return local;
    }

    // ---- path trace ------------------------------------------------------------------------
    // reflect (RayTracer.cpp:477-481): both branches are the same expression.
    static V3 reflect(const V3& v, const V3& n) { return v - n * 2 * dot(v, n); }
    // refract (RayTracer.cpp:430-448)
    static V3 refract(const V3& I, const V3& N, double eta_t, double eta_i) {
        double cosi = -std::max(-1.0, std::min(1.0, dot(I, N)));
        if (cosi < 0) return refract(I, -N, eta_i, eta_t);
        double eta = eta_i / eta_t;
        double k = 1 - eta * eta * (1 - cosi * cosi);
        if (k < 0) return V3(0, 0, 0);
        return I * eta + N * (eta * cosi - sqrt(k));
    }
    // fresnel (RayTracer.cpp:453-472)
    static double fresnel(const V3& I, const V3& N, double eta_t, double eta_i) {
        double cosi = std::clamp(dot(I, N), -1.0, 1.0);
        double etai = eta_i, etat = eta_t;
        if (cosi > 0) std::swap(etai, etat);
        double sint = etai / etat * sqrt(std::max(0.0, 1 - cosi * cosi));
        if (sint >= 1) return 1.0;
        double cost = sqrt(std::max(0.0, 1 - sint * sint));
        cosi = fabs(cosi);
        double Rs = ((etat * cosi) - (etai * cost)) / ((etat * cosi) + (etai * cost));
        double Rp = ((etai * cosi) - (etat * cost)) / ((etai * cosi) + (etat * cost));
        return (Rs * Rs + Rp * Rp) / 2.0;
    }
    // randomInHemisphere (RayTracer.cpp:398-424) with r1, r2 from the counter RNG
    static V3 hemisphere(const V3& n, double r1, double r2) {
        double sinTheta = sqrt(1 - r1 * r1);
        double phi = 2 * M_PI * r2;
        double x = cos(phi) * sinTheta;
        double y = r1;
        double z = sin(phi) * sinTheta;
        V3 tangent;
        if (fabs(n.x) > fabs(n.y)) tangent = normalize(V3(-n.z, 0, n.x));
        else tangent = normalize(V3(0, n.z, -n.y));
        V3 bitangent = cross(n, tangent);
        V3 dir = tangent * x + n * y + bitangent * z;
        return normalize(dir);
    }

    // estimateDirectLight (RayTracer.cpp:598-689) + AreaLight::sample (src/AreaLight.cpp:3-20)
    V3 direct_light(const Hit& h, const V3& view, const Rng& rng, uint32_t node, Counters& c) const {
        const rt_material& m = sc.mats[sc.prims[h.prim].material];
        V3 direct(0, 0, 0);
        for (size_t li = 0; li < sc.lights.size(); ++li) {
            const Light& L = sc.lights[li];
            if (L.type == RT_LIGHT_POINT) {
                V3 ldir = normalize(L.position - h.point);
                double distance = length(L.position - h.point);
                Ray shadow(h.point + h.normal * P.shadow_bias, ldir);
                Hit sh;
                if (sc.intersect(shadow, sh, c) && sh.t < distance) continue;
                double ndotl = std::max(0.0, dot(h.normal, ldir));
                V3 dbrdf = (V3(m.diffuse) * m.kd) / M_PI;
                if (m.texture >= 0) {
                    double u, v;
                    sc.prim_uv(h.prim, h.point, u, v);
                    dbrdf = sc.texture_color(m, u, v) * m.kd / M_PI;
                }
                V3 half = normalize(ldir + view);
                double ndoth = std::max(0.0, dot(h.normal, half));
                double sf = pow(ndoth, (double)m.specular_exponent);
                V3 sbrdf = V3(m.specular) * m.ks * ((m.specular_exponent + 2.0) / (2.0 * M_PI)) * sf;
                V3 brdf = dbrdf + sbrdf;
                direct = direct + brdf * L.intensity * ndotl;
            } else {
                V3 acc(0, 0, 0);
                for (int i = 0; i < P.light_samples; ++i) {
                    uint32_t dim = DIM_LIGHT0 + 2u * (uint32_t)(li * (size_t)P.light_samples + i);
                    double su = (rng.u(node, dim) - 0.5) * L.width;
                    double sv = (rng.u(node, dim + 1) - 0.5) * L.height;
                    V3 sp = L.position + L.u * su + L.v * sv;
                    V3 ldir = sp - h.point;
                    double distance = length(ldir);
                    ldir = ldir / distance;
                    double area = L.width * L.height;
                    double cosine = std::max(0.0, dot(L.normal, -ldir));
                    double pdf = (distance * distance) / (area * cosine);
                    Ray shadow(h.point + h.normal * P.shadow_bias, ldir);
                    Hit sh;
                    if (sc.intersect(shadow, sh, c) && sh.t < distance) continue;
                    double ndotl = std::max(0.0, dot(h.normal, ldir));
                    double ndotl_light = std::max(0.0, dot(L.normal, -ldir));
                    if (ndotl > 0 && ndotl_light > 0) {
                        V3 dbrdf = (V3(m.diffuse) * m.kd) / M_PI;
                        if (m.texture >= 0) {
                            double u, v;
                            sc.prim_uv(h.prim, h.point, u, v);
                            dbrdf = sc.texture_color(m, u, v) * m.kd / M_PI;
                        }
                        V3 half = normalize(ldir + view);
                        double ndoth = std::max(0.0, dot(h.normal, half));
                        double sf = pow(ndoth, (double)m.specular_exponent);
                        V3 sbrdf = V3(m.specular) * m.ks * ((m.specular_exponent + 2.0) / (2.0 * M_PI)) * sf;
                        V3 brdf = dbrdf + sbrdf;
                        acc = acc + brdf * L.intensity * ndotl * ndotl_light / pdf;
                    }
                }
                direct = direct + acc / (double)P.light_samples;
            }
        }
        return direct;
    }

    // traceRayPath (RayTracer.cpp:506-595); `node` is the RNG tree code of the hit this ray finds.
    V3 trace_path(const Ray& ray, int depth, const Rng& rng, uint32_t node, Counters& c) const {
        if (depth >= P.max_depth) return V3(0, 0, 0);
        Hit h;
        if (!sc.intersect(ray, h, c)) return sc.background;
        const rt_material& m = sc.mats[sc.prims[h.prim].material];
        V3 normal = h.normal;
        if (dot(ray.d, normal) > 0) normal = -normal;
        V3 albedo(m.diffuse);
        if (m.texture >= 0) {
            double u, v;
            sc.prim_uv(h.prim, h.point, u, v);
            albedo = sc.texture_color(m, u, v);
        }
        if (depth > 3) {
            double maxR = std::max(albedo.x, std::max(albedo.y, albedo.z));
            if (rng.u(node, DIM_RR) > maxR) return V3(0, 0, 0);
            albedo = albedo / maxR;
        }
        V3 direct = direct_light(h, -normalize(ray.d), rng, node, c);
        V3 indirect(0, 0, 0);
        if (m.is_reflective) {
            V3 rdir = normalize(reflect(normalize(ray.d), normal));
            Ray rr(h.point + normal * P.shadow_bias, rdir);
            V3 rc = trace_path(rr, depth + 1, rng, 2 * node, c);
            indirect = rc * m.reflectivity;
        } else if (m.is_refractive) {
            normal = h.normal;
            double eta_i = 1.0, eta_t = m.refractive_index;
            V3 incident = normalize(ray.d);
            bool entering = dot(incident, normal) < 0;
            if (!entering) {
                std::swap(eta_i, eta_t);
                normal = -normal;
            }
            double F = fresnel(incident, normal, eta_t, eta_i);
            V3 bias = normal * P.shadow_bias;
            V3 rdir = normalize(reflect(incident, normal));
            Ray rr(h.point + bias, rdir);
            V3 rc = trace_path(rr, depth + 1, rng, 2 * node, c);
            V3 tdir = normalize(refract(incident, normal, eta_t, eta_i));
            if (length(tdir) > 0.0) {  // NaN on total internal reflection -> false
                Ray tr(h.point - bias, tdir);
                V3 tc = trace_path(tr, depth + 1, rng, 2 * node + 1, c);
                indirect = rc * F + tc * (1.0 - F);
            } else {
                indirect = rc;
            }
        } else {
            V3 nd = hemisphere(normal, rng.u(node, DIM_HEMI_1), rng.u(node, DIM_HEMI_2));
            double cosTheta = std::max(0.0, dot(nd, normal));
            Ray nr(h.point + normal * P.shadow_bias, nd);
            indirect = trace_path(nr, depth + 1, rng, 2 * node, c) * (albedo / M_PI) * cosTheta;
        }
        return direct + indirect;
    }
};

inline double clamp01(double x) { return std::min(1.0, std::max(0.0, x)); }

Cam make_cam(const rt_camera* c) {
    Cam k;
    k.position = V3(c->position);
    k.lower_left = V3(c->lower_left);
    k.horizontal = V3(c->horizontal);
    k.vertical = V3(c->vertical);
    k.u = V3(c->u);
    k.v = V3(c->v);
    k.lens_radius = c->lens_radius;
    return k;
}

}  // namespace

static uint32_t* g_ray_map = nullptr;  // optional per-pixel Scene::intersect call counts (diagnostics)

extern "C" {

void oracle_set_ray_map(uint32_t* map) { g_ray_map = map; }

// RayTracer::render / renderPathTrace without the PPM write (RayTracer.cpp:206-246, 254-328).
// rgb: 3*W*H doubles (row 0 = bottom), hit_id: W*H or NULL, counters: {rays, boxes, prims} or NULL.
// accum (path trace only, may be NULL): un-normalised radiance sums before RayTracer.cpp:301.
int oracle_render(const rt_scene_desc* desc, const rt_camera* cam, const rt_params* P, double* rgb,
                  int32_t* hit_id, uint64_t* counters, double* accum, int n_threads) {
    Scene sc;
    load_scene(desc, sc);
    Cam C = make_cam(cam);
    Renderer R(sc, *P);
    const int W = P->width, H = P->height;
    uint64_t rays = 0, boxes = 0, prims = 0;
#ifdef _OPENMP
    if (n_threads > 0) omp_set_num_threads(n_threads);
#endif
    if (P->mode != RT_MODE_PATHTRACE) {
#pragma omp parallel for schedule(dynamic) reduction(+ : rays, boxes, prims)
        for (int j = 0; j < H; ++j) {
            Counters c;
            for (int i = 0; i < W; ++i) {
                double u = 1.0 - (double(i) / (W - 1));
                double v = double(j) / (H - 1);
                Ray ray = camera_ray(C, u, v, false, nullptr);
                int first = -1;
                uint64_t before = c.rays;
                V3 color = R.trace(ray, 0, c, &first);
                if (g_ray_map) g_ray_map[(size_t)j * W + i] = (uint32_t)(c.rays - before);
                color = tone_map(color, P->tonemap);
                color = color * P->exposure;
                double* px = rgb + ((size_t)j * W + i) * 3;
                px[0] = clamp01(color.x);
                px[1] = clamp01(color.y);
                px[2] = clamp01(color.z);
                if (hit_id) hit_id[(size_t)j * W + i] = first;
            }
            rays += c.rays; boxes += c.boxes; prims += c.prims;
        }
    } else {
        const int sq = static_cast<int>(std::sqrt(P->pixel_samples));
        const int s_begin = P->sample_begin;
        const int s_end = P->sample_end > 0 ? P->sample_end : sq * sq;
#pragma omp parallel for schedule(dynamic) reduction(+ : rays, boxes, prims)
        for (int j = 0; j < H; ++j) {
            Counters c;
            for (int i = 0; i < W; ++i) {
                V3 color(0, 0, 0);
                Rng rng;
                rng.key[0] = (uint32_t)P->seed;
                rng.key[1] = (uint32_t)(P->seed >> 32);
                rng.pixel = (uint32_t)((size_t)j * W + i);
                for (int sy = 0; sy < sq; ++sy) {
                    for (int sx = 0; sx < sq; ++sx) {
                        int s = sy * sq + sx;
                        if (s < s_begin || s >= s_end) continue;
                        rng.sample = (uint32_t)s;
                        const bool jitter = !(P->debug_flags & RT_DEBUG_NO_JITTER);
                        double r1 = (sx + (jitter ? rng.u(0, DIM_JITTER_X) : 0.0)) / sq;
                        double r2 = (sy + (jitter ? rng.u(0, DIM_JITTER_Y) : 0.0)) / sq;
                        double u = 1.0 - (double(i) + r1) / (W - 1);
                        double v = (double(j) + r2) / (H - 1);
                        Ray ray = camera_ray(C, u, v, true, &rng);
                        color = color + R.trace_path(ray, 0, rng, 1, c);
                    }
                }
                double* px = rgb + ((size_t)j * W + i) * 3;
                if (accum) {
                    double* ax = accum + ((size_t)j * W + i) * 3;
                    ax[0] = color.x; ax[1] = color.y; ax[2] = color.z;
                }
                color = color / (double)P->pixel_samples;
                color = tone_map(color, P->tonemap);
                color = color * P->exposure;
                color.x = pow(color.x, 1.0 / 2.2);
                color.y = pow(color.y, 1.0 / 2.2);
                color.z = pow(color.z, 1.0 / 2.2);
                px[0] = clamp01(color.x);
                px[1] = clamp01(color.y);
                px[2] = clamp01(color.z);
                if (hit_id) hit_id[(size_t)j * W + i] = -2;
            }
            rays += c.rays; boxes += c.boxes; prims += c.prims;
        }
    }
    if (counters) {
        counters[0] = rays;
        counters[1] = boxes;
        counters[2] = prims;
    }
    return 0;
}

// Scene::intersect on caller-supplied rays (directions are normalized by the Ray constructor,
// as everywhere in the reference).  t_out[k] = t or -1, prim_out[k] = ID or -1.
int oracle_trace_rays(const rt_scene_desc* desc, int64_t n, const double* origins, const double* dirs,
                      double* t_out, int32_t* prim_out, double* normal_out) {
    Scene sc;
    load_scene(desc, sc);
#pragma omp parallel for schedule(static)
    for (int64_t k = 0; k < n; ++k) {
        Counters c;
        Ray r(V3(origins + 3 * k), V3(dirs + 3 * k));
        Hit h;
        bool ok = sc.intersect(r, h, c);
        t_out[k] = ok ? h.t : -1.0;
        prim_out[k] = ok ? h.prim : -1;
        if (normal_out) {
            normal_out[3 * k] = ok ? h.normal.x : 0;
            normal_out[3 * k + 1] = ok ? h.normal.y : 0;
            normal_out[3 * k + 2] = ok ? h.normal.z : 0;
        }
    }
    return 0;
}

// BoundingBox::intersect on one box / one ray (direction used as given, NOT normalized, so that
// zero components and NaN cases can be driven exactly).
int oracle_box_hit(const double box[6], const double o[3], const double d[3]) {
    Box b{V3(box), V3(box + 3)};
    Ray r(V3(0, 0, 1), V3(0, 0, 1));
    r.o = V3(o);
    r.d = V3(d);
    return box_hit(b, r) ? 1 : 0;
}

// Number of BVHNode objects and the DFS (left-to-right) leaf order of primitive IDs.
int oracle_bvh_leaf_order(const rt_scene_desc* desc, int32_t* order_out, int32_t* n_nodes_out) {
    Scene sc;
    load_scene(desc, sc);
    int k = 0;
    struct Walk {
        static void go(const Node* n, int32_t* out, int& k) {
            if (n->leaf) { out[k++] = n->object; return; }
            if (n->left.node) go(n->left.node, out, k); else out[k++] = n->left.prim;
            if (n->right.node) go(n->right.node, out, k); else out[k++] = n->right.prim;
        }
    };
    if (sc.root) Walk::go(sc.root, order_out, k);
    if (n_nodes_out) *n_nodes_out = (int32_t)sc.pool.size();
    return k;
}

void oracle_tonemap(const double in[3], int tm, double out[3]) {
    V3 c = tone_map(V3(in), tm);
    out[0] = c.x; out[1] = c.y; out[2] = c.z;
}

// reflect / refract / fresnel / fresnelReflectance (RayTracer.cpp:430-504) on n rows of (I xyz, N xyz, eta_t, eta_i):
// out row = refract xyz, [length(normalize(refract)) > 0] (the transmission test of :572), reflect xyz, fresnel,
// fresnelReflectance(|I.N|, eta_t).  Checked against the reference's own functions (tests/golden/unit_vectors.npz).
void oracle_optics(int64_t n, const double* in, double* out) {
    for (int64_t k = 0; k < n; ++k) {
        const double* r = in + 8 * k;
        double* o = out + 9 * k;
        V3 I(r), N(r + 3);
        V3 t = Renderer::refract(I, N, r[6], r[7]);
        V3 tn = normalize(t);
        V3 rf = Renderer::reflect(I, N);
        o[0] = t.x; o[1] = t.y; o[2] = t.z;
        o[3] = length(tn) > 0.0 ? 1.0 : 0.0;
        o[4] = rf.x; o[5] = rf.y; o[6] = rf.z;
        o[7] = Renderer::fresnel(I, N, r[6], r[7]);
        o[8] = Renderer::schlick(std::fabs(dot(I, N)), r[6]);
    }
}

// estimateDirectLight (RayTracer.cpp:598-689) at the closest hit of n caller-supplied rays (directions normalized by the Ray
// constructor), view direction = -ray.direction.normalize() as traceRayPath passes it (:539): out row = (hit, direct xyz).
// Deterministic for point lights; area-light samples draw from the stream of pixel 0, stratum 0, node 1.
int oracle_direct_light(const rt_scene_desc* desc, const rt_params* P, int64_t n, const double* origins, const double* dirs, double* out) {
    Scene sc;
    load_scene(desc, sc);
    Renderer R(sc, *P);
    for (int64_t k = 0; k < n; ++k) {
        Counters c;
        Ray r(V3(origins + 3 * k), V3(dirs + 3 * k));
        Hit h;
        const bool ok = sc.intersect(r, h, c);
        Rng rng;
        rng.key[0] = (uint32_t)P->seed;
        rng.key[1] = (uint32_t)(P->seed >> 32);
        rng.pixel = 0;
        rng.sample = 0;
        const V3 d = ok ? R.direct_light(h, -normalize(r.d), rng, 1, c) : V3(0, 0, 0);
        out[4 * k] = ok ? 1.0 : 0.0;
        out[4 * k + 1] = d.x; out[4 * k + 2] = d.y; out[4 * k + 3] = d.z;
    }
    return 0;
}

// getUV of primitive prim[k] at point[k] (Sphere.cpp:50-57, Triangle.cpp:15-39, Cylinder.cpp:109-129), then
// Material::getTextureColor of that primitive's material at (u, v) (Material.cpp:69-86): out row = (u, v, r, g, b).
// uv_in != NULL: skip getUV and look the texture up at the given (u, v).
int oracle_uv_texture(const rt_scene_desc* desc, int64_t n, const int32_t* prim, const double* points, const double* uv_in, double* out) {
    Scene sc;
    load_scene(desc, sc);
    for (int64_t k = 0; k < n; ++k) {
        double u, v;
        if (uv_in) { u = uv_in[2 * k]; v = uv_in[2 * k + 1]; }
        else sc.prim_uv(prim[k], V3(points + 3 * k), u, v);
        const V3 c = sc.texture_color(sc.mats[sc.prims[prim[k]].material], u, v);
        out[5 * k] = u; out[5 * k + 1] = v; out[5 * k + 2] = c.x; out[5 * k + 3] = c.y; out[5 * k + 4] = c.z;
    }
    return 0;
}

void oracle_philox(const uint32_t ctr[4], const uint32_t key[2], uint32_t out[4]) { philox4x32_10(ctr, key, out); }

double oracle_uniform(uint64_t seed, uint32_t pixel, uint32_t sample, uint32_t node, uint32_t dim) {
    Rng r;
    r.key[0] = (uint32_t)seed;
    r.key[1] = (uint32_t)(seed >> 32);
    r.pixel = pixel;
    r.sample = sample;
    return r.u(node, dim);
}

int oracle_max_threads(void) {
#ifdef _OPENMP
    return omp_get_max_threads();
#else
    return 1;
#endif
}

}  // extern "C"
