// integration_check.cpp -- the binding of INTEGRATION.md section 2, as a real translation unit.
//
// TEST INFRASTRUCTURE, container only: `make -C oracle integration-check` compiles this file against the
// reference's own headers (/root/reference/include) and links it with the reference's classes (its sources are
// compiled where they lie, main renamed) and with libraytracer_amd.so.  It proves that the C-ABI in
// include/raytracer_amd.h can be fed from the reference's objects at its seam (src/RayTracer.cpp:146-149) with
// nothing but public members, and -- run on a GPU -- that the frame it returns is the frame the reference's own render()
// computes from the same objects.  The output goes to oracle/_ref/ (git-ignored and gpurun-ignored: nothing built from
// the reference leaves the build container).
//
// render_on_gpu() is what a maintainer of the reference would call instead of rayTracer.render(out) /
// rayTracer.renderPathTrace(out); the rest of main() stays as it is.
#include <cmath>
#include <cstdint>
#include <cstdio>
#include <cstdlib>
#include <iostream>
#include <string>
#include <vector>

#include "AreaLight.h"
#include "Camera.h"
#include "Cylinder.h"
#include "PointLight.h"
#include "RayTracer.h"
#include "Scene.h"
#include "Sphere.h"
#include "Triangle.h"
#include "raytracer_amd.h"

namespace {

void put3(double* dst, const Vector3& v) {
    dst[0] = v.x;
    dst[1] = v.y;
    dst[2] = v.z;
}

// Scene::objects / lights / backgroundColor / bvhRoot (include/Scene.h:17-21), the Camera constants
// (include/Camera.h:11-26) and the RayTracer scalars (include/RayTracer.h:35-47; the private ones are the values
// main() passed to the setters at :129-139) -> one frame in `buffer`, laid out as render() fills it (:235).
// Returns 0 or the failing rt_status; the message is rt_last_error().
int render_on_gpu(const Scene& scene, const Camera& cam, int W, int H, RayTracer::RenderMode mode,
                  RayTracer::ToneMapping toneMap, int maxDepth, int pixelSamples, int lightSamples, double exposure,
                  std::vector<std::vector<Vector3>>& buffer) {
    std::vector<int32_t> type, mat;
    std::vector<double> geom;
    std::vector<rt_material> mats;
    std::vector<rt_texture> texs;
    std::vector<std::vector<uint8_t>> texels;
    texels.reserve(scene.objects.size());  // rt_texture keeps pointers into these
    for (const auto& obj : scene.objects) {  // Scene::objects order = primitive ID
        double g[RT_GEOM_STRIDE] = {0};
        const Material* m = nullptr;
        if (auto s = dynamic_cast<const Sphere*>(obj.get())) {
            type.push_back(RT_SPHERE);
            put3(g, s->center);
            g[3] = s->radius;
            m = &s->material;
        } else if (auto t = dynamic_cast<const Triangle*>(obj.get())) {
            type.push_back(RT_TRIANGLE);
            put3(g, t->v0);
            put3(g + 3, t->v1);
            put3(g + 6, t->v2);
            m = &t->material;
        } else if (auto c = dynamic_cast<const Cylinder*>(obj.get())) {
            type.push_back(RT_CYLINDER);
            put3(g, c->baseCenter);
            put3(g + 3, c->axis);  // already normalized by the constructor; the library normalizes again (idempotent here)
            g[6] = c->radius;
            g[7] = c->height;
            m = &c->material;
        } else {
            std::cerr << "render_on_gpu: an Intersectable that is neither Sphere, Triangle nor Cylinder" << std::endl;
            return RT_ERR_INVALID;
        }
        geom.insert(geom.end(), g, g + RT_GEOM_STRIDE);
        rt_material rm;
        rm.ks = m->ks;
        rm.kd = m->kd;
        rm.reflectivity = m->reflectivity;
        rm.refractive_index = m->refractiveIndex;
        put3(rm.diffuse, m->diffuseColor);
        put3(rm.specular, m->specularColor);
        rm.specular_exponent = m->specularExponent;
        rm.is_reflective = m->isReflective ? 1 : 0;
        rm.is_refractive = m->isRefractive ? 1 : 0;
        rm.texture = -1;
        if (m->hasTexture && m->textureData && m->textureWidth > 0 && m->textureHeight > 0) {
            // Material::loadTexture stored byte / 255.0 per channel (src/Material.cpp:55-63): back to the bytes
            texels.emplace_back();
            for (const Vector3& c : *m->textureData)
                for (double ch : {c.x, c.y, c.z}) texels.back().push_back((uint8_t)std::lround(ch * 255.0));
            rt_texture tx;
            tx.width = m->textureWidth;
            tx.height = m->textureHeight;
            tx.rgb = texels.back().data();
            texs.push_back(tx);
            rm.texture = (int32_t)texs.size() - 1;
        }
        mat.push_back((int32_t)mats.size());
        mats.push_back(rm);
    }
    std::vector<rt_light> lights;
    for (const auto& l : scene.lights) {
        rt_light L = rt_light();
        put3(L.intensity, l->intensity);
        if (auto a = dynamic_cast<const AreaLight*>(l.get())) {
            L.type = RT_LIGHT_AREA;
            put3(L.position, a->position);
            put3(L.normal, a->normal);
            put3(L.u, a->uVec);
            put3(L.v, a->vVec);
            L.width = a->width;
            L.height = a->height;
        } else if (auto p = dynamic_cast<const PointLight*>(l.get())) {
            L.type = RT_LIGHT_POINT;
            put3(L.position, p->position);
        } else {
            continue;  // a bare Light cannot be constructed by parseLights (src/RayTracer.cpp:874-925)
        }
        lights.push_back(L);
    }
    rt_scene_desc d = rt_scene_desc();
    d.n_prims = (int32_t)type.size();
    d.n_materials = (int32_t)mats.size();
    d.n_textures = (int32_t)texs.size();
    d.n_lights = (int32_t)lights.size();
    d.prim_type = type.data();
    d.prim_material = mat.data();
    d.prim_geom = geom.data();
    d.materials = mats.data();
    d.textures = texs.data();
    d.lights = lights.data();
    put3(d.background, scene.backgroundColor);
    d.use_bvh = scene.bvhRoot ? 1 : 0;  // decides the tie rule (BVHNode.cpp:64-69 vs Scene.cpp:72-78)

    rt_camera c = rt_camera();
    put3(c.position, cam.position);
    put3(c.lower_left, cam.lowerLeftCorner);
    put3(c.horizontal, cam.horizontal);
    put3(c.vertical, cam.vertical);
    put3(c.u, cam.u);
    put3(c.v, cam.v);
    c.lens_radius = cam.lensRadius;

    rt_params p = rt_params();
    p.width = W;
    p.height = H;
    p.mode = mode == RayTracer::PHONG ? RT_MODE_PHONG : (mode == RayTracer::BINARY ? RT_MODE_BINARY : RT_MODE_PATHTRACE);
    p.tonemap = (int32_t)toneMap;  // same numbering
    p.max_depth = maxDepth;
    p.pixel_samples = pixelSamples;  // as SET, i.e. after the swap at src/RayTracer.cpp:138-139
    p.light_samples = lightSamples;
    p.exposure = exposure;
    p.shadow_bias = 1e-4;  // include/RayTracer.h:41
    p.seed = 0;
    p.band_rows = 8;
    p.band_rank = 0;
    p.band_world = 1;

    rt_context* ctx = nullptr;
    std::vector<double> rgb((size_t)W * H * 3);
    int rc = rt_create(&ctx, 0);
    if (rc == RT_OK) rc = rt_upload_scene(ctx, &d);
    if (rc == RT_OK) rc = rt_set_camera(ctx, &c);
    if (rc == RT_OK) rc = rt_set_params(ctx, &p);
    if (rc == RT_OK) rc = rt_render(ctx, rgb.data(), nullptr);
    if (ctx) rt_destroy(ctx);
    if (rc != RT_OK) return rc;
    buffer.assign(H, std::vector<Vector3>(W));
    for (int j = 0; j < H; ++j)  // row 0 = bottom row, as buffer[j][i] in render()
        for (int i = 0; i < W; ++i) {
            const double* px = &rgb[((size_t)j * W + i) * 3];
            buffer[j][i] = Vector3(px[0], px[1], px[2]);
        }
    return RT_OK;
}

}  // namespace

// Scenes built from the reference's own classes, rendered twice: through the binding on the GPU (written by the
// reference's own writer) and by the reference's own render() on the CPU, from the same objects.  The two P3 files must
// be identical byte for byte.  Exit status 0 = all identical, 2 = a difference, 1 = the library failed.
// (Linked in the build container; tests/test_gpu_integration_binary.py runs it wherever a gfx950 GPU and this binary are
// both present, and is skipped elsewhere.)
namespace {

std::string slurp(const std::string& path) {
    std::string out;
    if (FILE* f = std::fopen(path.c_str(), "rb")) {
        char buf[1 << 16];
        size_t n;
        while ((n = std::fread(buf, 1, sizeof buf, f)) > 0) out.append(buf, n);
        std::fclose(f);
    }
    return out;
}

// one comparison; returns 0 / 1 / 2 as above
int compare(const char* name, Scene& scene, Camera& camera, int W, int H, RayTracer::RenderMode mode, RayTracer::ToneMapping tm,
            int maxDepth, double exposure, const std::string& dir) {
    std::vector<std::vector<Vector3>> buffer;
    int rc = render_on_gpu(scene, camera, W, H, mode, tm, maxDepth, 16, 4, exposure, buffer);
    if (rc != RT_OK) {
        std::cerr << name << ": " << rt_last_error() << std::endl;
        return 1;
    }
    const std::string gpu_file = dir + "/" + name + "_gpu.ppm", ref_file = dir + "/" + name + "_ref.ppm";
    RayTracer rayTracer(&scene, &camera, W, H);  // the reference's main(), src/RayTracer.cpp:127-147
    rayTracer.setRenderMode(mode);
    rayTracer.setMaxDepth(maxDepth);
    rayTracer.setToneMap(tm);
    rayTracer.setExposure(exposure);
    rayTracer.writeImageToPPM(gpu_file, buffer);
    rayTracer.render(ref_file);
    const std::string a = slurp(gpu_file), b = slurp(ref_file);
    if (a.empty() || a != b) {
        size_t at = 0;
        while (at < a.size() && at < b.size() && a[at] == b[at]) ++at;
        std::cout << name << ": DIFFERENT (" << a.size() << " / " << b.size() << " bytes, first difference at byte " << at << ")" << std::endl;
        return 2;
    }
    std::cout << name << ": identical (" << a.size() << " bytes)" << std::endl;
    return 0;
}

}  // namespace

int main(int argc, char** argv) {
    const std::string dir = argc > 1 ? argv[1] : ".";
    const int W = 120, H = 80;
    int worst = 0;
    auto keep = [&](int rc) { worst = rc == 1 || worst == 1 ? 1 : (rc > worst ? rc : worst); };
    Material red(0.3, 0.7, 20, false, 0.0, false, 1.0, Vector3(0.8, 0.2, 0.2), Vector3(1, 1, 1));
    Material mirror(0.5, 0.4, 50, true, 0.6, false, 1.0, Vector3(0.9, 0.9, 0.9), Vector3(1, 1, 1));
    Material glass(0.2, 0.1, 80, false, 0.0, true, 1.5, Vector3(0.9, 0.9, 1.0), Vector3(1, 1, 1));
    Material green(0.1, 0.9, 5, false, 0.0, false, 1.0, Vector3(0.2, 0.7, 0.3), Vector3(1, 1, 1));
    Camera camera(Vector3(0, 0, 0), Vector3(0, 0, 1), Vector3(0, 1, 0), 45.0, double(W) / H);
    {   // every primitive class, a point light, the BVH's tie rule
        Scene scene(Vector3(0.2, 0.3, 0.4));
        scene.addObject(std::make_shared<Sphere>(Vector3(0, 0, 4), 1.0, red));
        scene.addObject(std::make_shared<Triangle>(Vector3(-3, -1, 1), Vector3(3, -1, 1), Vector3(0, -1, 8), green));
        scene.addObject(std::make_shared<Cylinder>(Vector3(1.5, -1, 4), Vector3(0, 1, 0), 0.4, 1.5, red));
        scene.addLight(std::make_shared<PointLight>(Vector3(3, 5, 0), Vector3(1, 1, 1)));
        scene.buildBVH();
        keep(compare("phong_bvh", scene, camera, W, H, RayTracer::PHONG, RayTracer::NONE, 5, 1.0, dir));
        keep(compare("binary_bvh", scene, camera, W, H, RayTracer::BINARY, RayTracer::NONE, 5, 1.0, dir));
    }
    {   // mirror and glass bounces, two lights, no BVH (the linear scan's tie rule), a tone map
        Scene scene(Vector3(0.25, 0.25, 0.3));
        scene.addObject(std::make_shared<Sphere>(Vector3(-1.2, 0, 5), 1.0, mirror));
        scene.addObject(std::make_shared<Sphere>(Vector3(1.0, -0.2, 3.5), 0.8, glass));
        scene.addObject(std::make_shared<Cylinder>(Vector3(0.2, -1, 6), Vector3(0.3, 1, 0.1), 0.5, 2.0, red));
        scene.addObject(std::make_shared<Triangle>(Vector3(-6, -1, 0), Vector3(6, -1, 0), Vector3(0, -1, 14), green));
        scene.addLight(std::make_shared<PointLight>(Vector3(3, 5, 0), Vector3(0.8, 0.8, 0.8)));
        scene.addLight(std::make_shared<PointLight>(Vector3(-4, 3, 2), Vector3(0.4, 0.4, 0.5)));
        keep(compare("whitted_linear", scene, camera, W, H, RayTracer::PHONG, RayTracer::REINHARD, 6, 1.3, dir));
        scene.buildBVH();
        keep(compare("whitted_bvh", scene, camera, W, H, RayTracer::PHONG, RayTracer::UNCHARTED2, 6, 1.0, dir));
    }
    return worst;
}
