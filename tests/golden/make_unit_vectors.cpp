// make_unit_vectors.cpp -- per-function known answers from the REFERENCE'S OWN OBJECTS.
//
// Container only.  tests/golden/make_unit_vectors.py compiles this driver against the reference's headers and
// links it with the reference's translation units, compiled where they lie under /root/reference (main renamed),
// runs it and packs its output into tests/golden/unit_vectors.npz -- inputs and the reference's outputs, data
// only.  The oracle (oracle/oracle.cpp) and the GPU (rt_debug_trace_rays / rt_debug_optics) are then checked
// against these vectors on the edge cases the reference relies on:
//   BoundingBox::intersect  zero direction components (division by +-0), origins on slab planes (0/0 = NaN);
//   Sphere / Triangle / Cylinder::intersect  grazing rays, origins inside, d.axis = 0 (cap division by zero);
//   Scene::intersect with a BVH on the same adversarial rays (gate boxes, tie rule);
//   refract / fresnel / fresnelReflectance / reflect  incl. total internal reflection and the flip recursion;
//   toneMap  all four operators;
//   traceRayPath / estimateDirectLight  (private members: this driver, and only it, includes RayTracer.h with
//     `private` spelled `public`; the reference's own translation units are compiled unmodified) on a scene of mirrors
//     and glass under a point light, recursion depth <= 3: no Russian roulette, no hemisphere sample, no area-light
//     sample -- fully deterministic (RayTracer.cpp:506-637), total internal reflection through the NaN of :572 included;
//   getUV of the three classes and Material::getTextureColor at texel boundaries.
#include <cmath>
#include <cstdint>
#include <cstdio>
#include <cstring>
#include <memory>
#include <string>
#include <vector>

#include "BoundingBox.h"
#include "Camera.h"
#include "Cylinder.h"
#include "PointLight.h"
#define private public  // (this driver only: traceRayPath and estimateDirectLight are private members)
#include "RayTracer.h"
#undef private
#include "Scene.h"
#include "Sphere.h"
#include "Triangle.h"

// free functions of src/RayTracer.cpp (external linkage there)
Vector3 toneMap(Vector3 color, RayTracer::ToneMapping toneMapping);
Vector3 refract(const Vector3& I, const Vector3& N, double eta_t, double eta_i);
double fresnel(const Vector3& I, const Vector3& N, double eta_t, double eta_i);
Vector3 reflect(const Vector3& v, const Vector3& n);
double fresnelReflectance(double cosTheta, double refractiveIndex);

namespace {

struct Lcg {  // deterministic inputs (not the reference's RNG: these are test inputs)
    uint64_t s;
    explicit Lcg(uint64_t seed) : s(seed) {}
    double u() {
        s = s * 6364136223846793005ull + 1442695040888963407ull;
        return (double)(s >> 11) * 0x1.0p-53;
    }
    double range(double a, double b) { return a + (b - a) * u(); }
    int pick(int n) { return (int)(u() * n) % n; }
};

FILE* g_out = nullptr;
void emit(const char* name, const std::vector<double>& v, int cols) {
    uint32_t len = (uint32_t)std::strlen(name), rows = (uint32_t)(v.size() / cols), c = (uint32_t)cols;
    std::fwrite(&len, 4, 1, g_out);
    std::fwrite(name, 1, len, g_out);
    std::fwrite(&rows, 4, 1, g_out);
    std::fwrite(&c, 4, 1, g_out);
    std::fwrite(v.data(), 8, v.size(), g_out);
}
void push3(std::vector<double>& v, const Vector3& a) {
    v.push_back(a.x);
    v.push_back(a.y);
    v.push_back(a.z);
}

// adversarial rays against the box [lo, hi]: axis-parallel directions (zero components), origins exactly on slab
// planes / corners / inside, plus random ones
void adversarial_rays(Lcg& g, const Vector3& lo, const Vector3& hi, int n, std::vector<Vector3>& org, std::vector<Vector3>& dir) {
    const double lo_[3] = {lo.x, lo.y, lo.z}, hi_[3] = {hi.x, hi.y, hi.z};
    for (int k = 0; k < n; ++k) {
        double o[3], d[3];
        for (int a = 0; a < 3; ++a) {
            const double ext = hi_[a] - lo_[a];
            switch (g.pick(6)) {
                case 0: o[a] = lo_[a]; break;                       // on the lower plane
                case 1: o[a] = hi_[a]; break;                       // on the upper plane
                case 2: o[a] = 0.5 * (lo_[a] + hi_[a]); break;      // centre
                case 3: o[a] = g.range(lo_[a], hi_[a]); break;      // inside
                default: o[a] = g.range(lo_[a] - 2 * ext - 1, hi_[a] + 2 * ext + 1);
            }
            switch (g.pick(5)) {
                case 0: d[a] = 0.0; break;
                case 1: d[a] = -0.0; break;
                case 2: d[a] = g.pick(2) ? 1.0 : -1.0; break;
                default: d[a] = g.range(-1, 1);
            }
        }
        if (d[0] == 0 && d[1] == 0 && d[2] == 0) d[g.pick(3)] = g.pick(2) ? 1.0 : -1.0;
        org.push_back(Vector3(o[0], o[1], o[2]));
        dir.push_back(Vector3(d[0], d[1], d[2]));
    }
}

void primitive_vectors(const char* name, const Intersectable& prim, const std::vector<Vector3>& org, const std::vector<Vector3>& dir) {
    std::vector<double> in, out;
    for (size_t k = 0; k < org.size(); ++k) {
        Ray ray(org[k], dir[k]);  // normalizes the direction (src/Ray.cpp:5-6)
        HitRecord h;
        bool hit = prim.intersect(ray, h);
        push3(in, org[k]);
        push3(in, dir[k]);
        out.push_back(hit ? 1.0 : 0.0);
        out.push_back(hit ? h.t : 0.0);
        push3(out, hit ? h.point : Vector3(0, 0, 0));
        push3(out, hit ? h.normal : Vector3(0, 0, 0));
    }
    emit((std::string(name) + "_in").c_str(), in, 6);
    emit((std::string(name) + "_out").c_str(), out, 8);
}

}  // namespace

int main(int argc, char** argv) {
    g_out = std::fopen(argc > 1 ? argv[1] : "unit_vectors.bin", "wb");
    if (!g_out) return 1;
    Lcg g(20261111);
    Material mat(0.5, 0.5, 10, false, 0.0, false, 1.0, Vector3(1, 1, 1), Vector3(1, 1, 1));

    // ---- BoundingBox::intersect (src/BoundingBox.cpp:24-62) ----------------------------------------------
    {
        std::vector<double> in, out;
        for (int b = 0; b < 40; ++b) {
            Vector3 c(g.range(-3, 3), g.range(-3, 3), g.range(-3, 3));
            Vector3 e(g.range(0.1, 2), g.range(0.1, 2), b % 5 == 0 ? 0.0 : g.range(0.1, 2));  // flat boxes too (triangles)
            BoundingBox box(c - e, c + e);
            std::vector<Vector3> org, dir;
            adversarial_rays(g, box.min, box.max, 120, org, dir);
            for (size_t k = 0; k < org.size(); ++k) {
                Ray ray(org[k], dir[k]);
                double tn = 0, tf = 0;
                bool hit = box.intersect(ray, tn, tf);
                push3(in, box.min);
                push3(in, box.max);
                push3(in, ray.origin);
                push3(in, ray.direction);  // AS SEEN by BoundingBox::intersect (normalized by the Ray constructor)
                out.push_back(hit ? 1.0 : 0.0);
            }
        }
        emit("box_in", in, 12);
        emit("box_out", out, 1);
    }

    // ---- primitives (src/Sphere.cpp:14-48, src/Triangle.cpp:42-80, src/Cylinder.cpp:13-107) -----------------
    Sphere sphere(Vector3(0.5, -0.25, 3.0), 1.25, mat);
    Triangle triangle(Vector3(-1, -0.5, 2), Vector3(1.5, -0.5, 2.5), Vector3(0.25, 1.5, 4), mat);
    Cylinder cyl_y(Vector3(0.3, -0.5, 1.0), Vector3(0, 1, 0), 0.25, 1.0, mat);
    Cylinder cyl_x(Vector3(-1.0, 0.5, 2.0), Vector3(1, 0, 0), 0.5, 2.0, mat);
    Cylinder cyl_d(Vector3(0.0, 0.0, 3.0), Vector3(1, 2, -0.5), 0.4, 1.5, mat);
    {
        std::vector<Vector3> org, dir;
        adversarial_rays(g, Vector3(-0.75, -1.5, 1.75), Vector3(1.75, 1.0, 4.25), 1500, org, dir);
        // grazing: tangent rays and rays from inside the sphere
        for (int k = 0; k < 300; ++k) {
            Vector3 o(g.range(-4, 4), g.range(-4, 4), g.range(-2, 1));
            Vector3 tangent_point = sphere.center + Vector3(g.range(-1, 1), g.range(-1, 1), g.range(-1, 1)).normalize() * sphere.radius;
            org.push_back(o);
            dir.push_back(tangent_point - o);
            org.push_back(sphere.center + Vector3(g.range(-0.5, 0.5), g.range(-0.5, 0.5), g.range(-0.5, 0.5)));
            dir.push_back(Vector3(g.range(-1, 1), g.range(-1, 1), g.range(-1, 1)));
        }
        primitive_vectors("sphere", sphere, org, dir);
    }
    {
        std::vector<Vector3> org, dir;
        adversarial_rays(g, Vector3(-1, -0.5, 2), Vector3(1.5, 1.5, 4), 1500, org, dir);
        for (int k = 0; k < 300; ++k) {  // through edges / vertices, in the plane, from the plane
            double a = g.u(), b = g.u() * (1 - a);
            if (k % 3 == 0) b = 0.0;
            if (k % 7 == 0) { a = 1.0; b = 0.0; }
            Vector3 p = triangle.v0 + (triangle.v1 - triangle.v0) * a + (triangle.v2 - triangle.v0) * b;
            Vector3 o(g.range(-3, 3), g.range(-3, 3), g.range(-1, 1));
            org.push_back(o);
            dir.push_back(p - o);
            org.push_back(p);
            dir.push_back(Vector3(g.range(-1, 1), g.range(-1, 1), g.range(-1, 1)));
        }
        primitive_vectors("triangle", triangle, org, dir);
    }
    struct { const char* name; const Cylinder* c; } cyls[3] = {{"cylinder_y", &cyl_y}, {"cylinder_x", &cyl_x}, {"cylinder_d", &cyl_d}};
    for (auto& cc : cyls) {
        const Cylinder& c = *cc.c;
        std::vector<Vector3> org, dir;
        Vector3 top = c.baseCenter + c.axis * c.height;
        Vector3 lo(std::min(c.baseCenter.x, top.x) - c.radius, std::min(c.baseCenter.y, top.y) - c.radius, std::min(c.baseCenter.z, top.z) - c.radius);
        Vector3 hi(std::max(c.baseCenter.x, top.x) + c.radius, std::max(c.baseCenter.y, top.y) + c.radius, std::max(c.baseCenter.z, top.z) + c.radius);
        adversarial_rays(g, lo, hi, 1500, org, dir);
        for (int k = 0; k < 300; ++k) {
            // perpendicular to the axis (d.axis = 0: the cap tests divide by zero), parallel to it, from inside
            Vector3 r(g.range(-1, 1), g.range(-1, 1), g.range(-1, 1));
            Vector3 perp = (r - c.axis * r.dot(c.axis));
            Vector3 o(g.range(lo.x - 2, hi.x + 2), g.range(lo.y - 2, hi.y + 2), g.range(lo.z - 2, hi.z + 2));
            org.push_back(o);
            dir.push_back(perp);
            org.push_back(o);
            dir.push_back(k % 2 ? c.axis : -c.axis);
            org.push_back(c.baseCenter + c.axis * (c.height * g.u()) + perp.normalize() * (c.radius * g.u()));
            dir.push_back(r);
        }
        primitive_vectors(cc.name, c, org, dir);
    }

    // ---- Scene::intersect with and without the BVH (src/Scene.cpp:63-81, src/BVHNode.cpp:47-81) -------------
    // five primitives tagged by material.ks = index; cyl_x's reference box assumes a +Y axis (Cylinder.cpp:131-137)
    for (int bvh = 0; bvh < 2; ++bvh) {
        Scene scene(Vector3(0, 0, 0));
        const Intersectable* prims[5] = {&sphere, &triangle, &cyl_y, &cyl_x, &cyl_d};
        for (int i = 0; i < 5; ++i) {
            Material tagged(double(i), 0.5, 10, false, 0.0, false, 1.0, Vector3(1, 1, 1), Vector3(1, 1, 1));
            if (i == 0) scene.addObject(std::make_shared<Sphere>(sphere.center, sphere.radius, tagged));
            else if (i == 1) scene.addObject(std::make_shared<Triangle>(triangle.v0, triangle.v1, triangle.v2, tagged));
            else {
                const Cylinder* c = static_cast<const Cylinder*>(prims[i]);
                scene.addObject(std::make_shared<Cylinder>(c->baseCenter, c->axis, c->radius, c->height, tagged));
            }
        }
        if (bvh) scene.buildBVH();
        Lcg gs(777);
        std::vector<Vector3> org, dir;
        adversarial_rays(gs, Vector3(-1.5, -1.5, 1.0), Vector3(2.0, 2.0, 4.5), 4000, org, dir);
        std::vector<double> in, out;
        for (size_t k = 0; k < org.size(); ++k) {
            Ray ray(org[k], dir[k]);
            HitRecord h;
            bool hit = scene.intersect(ray, h);
            push3(in, org[k]);
            push3(in, dir[k]);
            out.push_back(hit ? h.material.ks : -1.0);
            out.push_back(hit ? h.t : 0.0);
            push3(out, hit ? h.normal : Vector3(0, 0, 0));
        }
        emit(bvh ? "scene_bvh_in" : "scene_linear_in", in, 6);
        emit(bvh ? "scene_bvh_out" : "scene_linear_out", out, 5);
    }

    // ---- reflect / refract / fresnel / fresnelReflectance (src/RayTracer.cpp:430-504) ------------------------
    {
        std::vector<double> in, out;
        for (int k = 0; k < 4000; ++k) {
            Vector3 I = Vector3(g.range(-1, 1), g.range(-1, 1), g.range(-1, 1)).normalize();
            Vector3 N = Vector3(g.range(-1, 1), g.range(-1, 1), g.range(-1, 1)).normalize();
            if (k % 10 == 0) N = -I;                  // head-on
            if (k % 10 == 1) N = I;                   // from inside, head-on
            if (k % 10 == 2) N = (N - I * N.dot(I)).normalize();  // grazing: I.N ~ 0
            double eta_t = k % 4 == 0 ? 1.0 : g.range(1.0, 2.5), eta_i = k % 3 == 0 ? 1.0 : g.range(1.0, 2.5);
            Vector3 t = refract(I, N, eta_t, eta_i);
            Vector3 tn = t.normalize();               // what traceRayPath does next (:571): NaN on total internal reflection
            Vector3 r = reflect(I, N);
            push3(in, I);
            push3(in, N);
            in.push_back(eta_t);
            in.push_back(eta_i);
            push3(out, t);
            out.push_back(tn.length() > 0.0 ? 1.0 : 0.0);  // the test of :572
            push3(out, r);
            out.push_back(fresnel(I, N, eta_t, eta_i));
            out.push_back(fresnelReflectance(std::fabs(I.dot(N)), eta_t));
        }
        emit("optics_in", in, 8);
        emit("optics_out", out, 9);
    }

    // ---- toneMap (src/RayTracer.cpp:161-202) -----------------------------------------------------------------
    {
        std::vector<double> in, out;
        for (int k = 0; k < 1000; ++k) {
            Vector3 c(g.range(0, 1) * std::exp(g.range(-6, 4)), g.range(0, 1) * std::exp(g.range(-6, 4)), g.range(0, 3));
            if (k == 0) c = Vector3(0, 0, 0);
            push3(in, c);
            for (int tm = 0; tm < 4; ++tm) push3(out, toneMap(c, (RayTracer::ToneMapping)tm));
        }
        emit("tonemap_in", in, 3);
        emit("tonemap_out", out, 12);
    }
    // ---- traceRayPath / estimateDirectLight, deterministic part (src/RayTracer.cpp:506-637) -------------------
    // The scene is emitted as data (one row per primitive: type, material, 12 geometry values as rt_scene_desc takes them)
    // so that the tests rebuild it without a scene file.
    {
        const Vector3 bgc(0.53, 0.8, 0.92);
        Scene scene(bgc);
        struct M { double ks, kd; int e; bool refl; double r; bool refr; double ior; Vector3 dc, sc; };
        const M mats[5] = {{0.1, 0.9, 20, false, 1.0, true, 1.5, Vector3(0.8, 0.5, 0.5), Vector3(1, 1, 1)},    // glass
                           {0.3, 0.6, 10, true, 0.8, false, 1.0, Vector3(0.5, 0.5, 0.8), Vector3(1, 1, 1)},   // mirror
                           {0.2, 0.7, 5, true, 0.6, true, 1.3, Vector3(0.5, 0.8, 0.5), Vector3(1, 0.9, 0.8)}, // both flags: the mirror branch wins (:543)
                           {0.1, 0.9, 20, false, 1.0, true, 2.4, Vector3(0.9, 0.9, 0.9), Vector3(1, 1, 1)},   // dense glass: total internal reflection
                           {0.4, 0.5, 30, true, 1.0, false, 1.0, Vector3(0.7, 0.7, 0.3), Vector3(1, 1, 1)}};  // floor mirror
        auto mat_of = [&](int k) { const M& m = mats[k]; return Material(m.ks, m.kd, m.e, m.refl, m.r, m.refr, m.ior, m.dc, m.sc); };
        std::vector<double> prims, matrows;
        for (const M& m : mats) {
            const double row[13] = {m.ks, m.kd, (double)m.e, m.refl ? 1.0 : 0.0, m.r, m.refr ? 1.0 : 0.0, m.ior, m.dc.x, m.dc.y, m.dc.z, m.sc.x, m.sc.y, m.sc.z};
            matrows.insert(matrows.end(), row, row + 13);
        }
        auto add_sphere = [&](Vector3 c, double r, int m) {
            scene.addObject(std::make_shared<Sphere>(c, r, mat_of(m)));
            const double row[14] = {0, (double)m, c.x, c.y, c.z, r, 0, 0, 0, 0, 0, 0, 0, 0};
            prims.insert(prims.end(), row, row + 14);
        };
        auto add_tri = [&](Vector3 a, Vector3 b, Vector3 c, int m) {
            scene.addObject(std::make_shared<Triangle>(a, b, c, mat_of(m)));
            const double row[14] = {1, (double)m, a.x, a.y, a.z, b.x, b.y, b.z, c.x, c.y, c.z, 0, 0, 0};
            prims.insert(prims.end(), row, row + 14);
        };
        auto add_cyl = [&](Vector3 base, Vector3 axis, double r, double h, int m) {  // constructor arguments (the constructor normalizes the axis)
            scene.addObject(std::make_shared<Cylinder>(base, axis, r, h, mat_of(m)));
            const double row[14] = {2, (double)m, base.x, base.y, base.z, axis.x, axis.y, axis.z, r, h, 0, 0, 0, 0};
            prims.insert(prims.end(), row, row + 14);
        };
        add_sphere(Vector3(-0.35, -0.2, 1.0), 0.3, 0);
        add_sphere(Vector3(0.45, 0.1, 1.6), 0.35, 1);
        add_sphere(Vector3(-0.9, 0.0, 1.8), 0.25, 3);
        add_cyl(Vector3(0.3, -0.5, 0.8), Vector3(0, 1, 0), 0.2, 0.5, 2);
        add_cyl(Vector3(-0.2, 0.3, 1.9), Vector3(1, 0.5, 0.2), 0.15, 0.6, 0);
        add_tri(Vector3(-2, -0.5, 3), Vector3(2, -0.5, 3), Vector3(2, -0.5, -1), 4);
        add_tri(Vector3(-2, -0.5, -1), Vector3(-2, -0.5, 3), Vector3(2, -0.5, -1), 4);
        add_tri(Vector3(-1.5, -0.5, 2.6), Vector3(1.5, -0.5, 2.6), Vector3(0, 1.8, 2.8), 1);
        const Vector3 lpos(0.5, 2.5, -0.5), lint(1.2, 1.1, 0.9);
        scene.addLight(std::make_shared<PointLight>(lpos, lint));
        const int W = 48, H = 32, max_depth = 4;  // depths 0 .. 3: Russian roulette starts beyond (:530)
        const Vector3 cpos(-2, 1, 0), clook(0.0, -0.1, 1.0), cup(0, 1, 0);
        const double fov = 45.0, aspect = double(W) / H, focus = 2.0;
        for (int bvh = 0; bvh < 2; ++bvh) {
            if (bvh) scene.buildBVH();
            Camera camera(cpos, clook, cup, fov, aspect, 0.0, focus);
            RayTracer tracer(&scene, &camera, W, H);
            tracer.setMaxDepth(max_depth);
            tracer.setRenderMode(RayTracer::PATH_TRACE);
            tracer.setLightSample(4);
            std::vector<double> out, direct;
            for (int j = 0; j < H; ++j)
                for (int i = 0; i < W; ++i) {
                    const double u = 1.0 - double(i) / (W - 1), v = double(j) / (H - 1);  // render()'s rays (:217-218)
                    Ray ray = camera.getRay(u, v);
                    push3(out, tracer.traceRayPath(ray, 0));
                    HitRecord h;
                    const bool hit = scene.intersect(ray, h);
                    direct.push_back(hit ? 1.0 : 0.0);
                    push3(direct, hit ? tracer.estimateDirectLight(h, -ray.direction.normalize()) : Vector3(0, 0, 0));
                }
            emit(bvh ? "pt_trace_bvh_out" : "pt_trace_linear_out", out, 3);
            emit(bvh ? "pt_direct_bvh_out" : "pt_direct_linear_out", direct, 4);
        }
        emit("pt_prims", prims, 14);
        emit("pt_materials", matrows, 13);
        const double setup[20] = {(double)W, (double)H, (double)max_depth, fov, focus, cpos.x, cpos.y, cpos.z, clook.x, clook.y, clook.z,
                                  lpos.x, lpos.y, lpos.z, lint.x, lint.y, lint.z, bgc.x, bgc.y, bgc.z};
        emit("pt_setup", std::vector<double>(setup, setup + 20), 20);
    }

    // ---- getUV (src/Sphere.cpp:50-57, src/Triangle.cpp:15-39, src/Cylinder.cpp:109-129), getTextureColor (src/Material.cpp:69-86) ----
    {
        const Intersectable* prims[5] = {&sphere, &triangle, &cyl_y, &cyl_x, &cyl_d};
        const char* names[5] = {"uv_sphere", "uv_triangle", "uv_cylinder_y", "uv_cylinder_x", "uv_cylinder_d"};
        for (int p = 0; p < 5; ++p) {
            std::vector<double> rows;
            Lcg gu(4242 + p);
            int found = 0;
            for (int k = 0; k < 200000 && found < 600; ++k) {
                Ray ray(Vector3(gu.range(-4, 4), gu.range(-4, 4), gu.range(-3, 0.5)), Vector3(gu.range(-1, 1), gu.range(-1, 1), gu.range(0.2, 1)));
                HitRecord h;
                if (!prims[p]->intersect(ray, h) || !h.getUV) continue;
                double u = 0, v = 0;
                h.getUV(h.point, u, v);
                push3(rows, h.point);
                rows.push_back(u);
                rows.push_back(v);
                ++found;
            }
            emit(names[p], rows, 5);
        }
        Material textured(0.5, 0.5, 10, false, 0.0, false, 1.0, Vector3(0.1, 0.2, 0.3), Vector3(1, 1, 1), true, "texture/checkerboard.ppm");
        std::vector<double> rows;
        Lcg gt(99);
        const int TW = textured.textureWidth, TH = textured.textureHeight;
        for (int k = 0; k < 3000; ++k) {
            double u = gt.range(-1.5, 2.5), v = gt.range(-1.5, 2.5);
            if (k % 3 == 0 && TW > 0) u = double(gt.pick(TW + 1)) / TW + (k % 2 ? 1e-17 : -1e-17) * (k % 5);   // texel boundaries
            if (k % 4 == 0 && TH > 0) v = 1.0 - double(gt.pick(TH + 1)) / TH;
            if (k == 1) { u = 0.0; v = 0.0; }
            if (k == 2) { u = 1.0; v = 1.0; }
            const Vector3 c = textured.getTextureColor(u, v);
            const double row[5] = {u, v, c.x, c.y, c.z};
            rows.insert(rows.end(), row, row + 5);
        }
        emit("texture_checkerboard", rows, 5);
        const double dims[2] = {(double)TW, (double)TH};
        emit("texture_checkerboard_size", std::vector<double>(dims, dims + 2), 2);
    }
    std::fclose(g_out);
    return 0;
}
