// host_scene.cpp -- the host side of the seam: everything the reference's main() does before
// rayTracer.render()/renderPathTrace() (RayTracer.cpp:52-143) and the PPM writer after it
// (RayTracer.cpp:352-367), restated from scratch.  No HIP in this file.
//
// Build with -ffp-contract=off: every double computed here (camera basis, cylinder transform,
// triangle normals, BVH boxes and centres) must have the bits the reference's x86-64 build
// produces, because the BVH topology and the primary-hit IDs depend on them.
#include <algorithm>
#include <cmath>
#include <cstdio>
#include <cstring>
#include <fstream>
#include <iostream>
#include <map>
#include <memory>
#include <sstream>
#include <string>
#include <vector>

#include "mini_json.hpp"
#include "rt_internal.hpp"

#ifndef M_PI
#define M_PI 3.14159265358979323846
#endif

namespace rt {

static thread_local std::string g_last_error;

int set_error(int code, const char* fmt, ...) {
    char buf[1024];
    va_list ap;
    va_start(ap, fmt);
    vsnprintf(buf, sizeof buf, fmt, ap);
    va_end(ap);
    g_last_error = buf;
    return code;
}

// --- Vector3 arithmetic in the reference's operation order (src/Vector3.cpp) -----------------
struct H3 {
    double x, y, z;
};
static inline H3 h3(const double* p) { return H3{p[0], p[1], p[2]}; }
static inline H3 add(H3 a, H3 b) { return H3{a.x + b.x, a.y + b.y, a.z + b.z}; }
static inline H3 sub(H3 a, H3 b) { return H3{a.x - b.x, a.y - b.y, a.z - b.z}; }
static inline H3 mul(H3 a, double s) { return H3{a.x * s, a.y * s, a.z * s}; }
static inline H3 div(H3 a, double s) { return H3{a.x / s, a.y / s, a.z / s}; }
static inline double dot(H3 a, H3 b) { return a.x * b.x + a.y * b.y + a.z * b.z; }
static inline H3 cross(H3 a, H3 b) { return H3{a.y * b.z - a.z * b.y, a.z * b.x - a.x * b.z, a.x * b.y - a.y * b.x}; }
static inline double length(H3 a) { return std::sqrt(a.x * a.x + a.y * a.y + a.z * a.z); }
static inline H3 normalize(H3 a) {
    double len = length(a);
    return H3{a.x / len, a.y / len, a.z / len};
}
static inline void put(double* p, H3 a) {
    p[0] = a.x;
    p[1] = a.y;
    p[2] = a.z;
}

void triangle_normal(const double* v0, const double* v1, const double* v2, double n[3]) {
    H3 a = h3(v0);
    H3 nn = normalize(cross(sub(h3(v1), a), sub(h3(v2), a)));
    if (dot(nn, a) > 0) nn = H3{-nn.x, -nn.y, -nn.z};
    put(n, nn);
}

void prim_bounds(const rt_scene_desc* s, int i, double box[6]) {
    const double* g = s->prim_geom + (size_t)i * RT_GEOM_STRIDE;
    switch (s->prim_type[i]) {
        case RT_SPHERE: {  // src/Sphere.cpp:60-63
            double r = g[3];
            for (int k = 0; k < 3; ++k) {
                box[k] = g[k] - r;
                box[3 + k] = g[k] + r;
            }
            break;
        }
        case RT_TRIANGLE: {  // src/Triangle.cpp:82-94
            for (int k = 0; k < 3; ++k) {
                box[k] = std::min({g[k], g[3 + k], g[6 + k]});
                box[3 + k] = std::max({g[k], g[3 + k], g[6 + k]});
            }
            break;
        }
        default: {  // src/Cylinder.cpp:131-137: base - r, (base + r) + (0, height, 0)
            double r = g[6], h = g[7];
            const double hv[3] = {0.0, h, 0.0};
            for (int k = 0; k < 3; ++k) {
                box[k] = g[k] - r;
                box[3 + k] = (g[k] + r) + hv[k];
            }
            break;
        }
    }
}

namespace {
struct Builder {
    const rt_scene_desc* s;
    std::vector<double> pbox;     // [n][6]
    std::vector<double> pcenter;  // [n][3]  (min + max) * 0.5, src/BoundingBox.cpp:64-66
    FlatBVH* out;

    int build(const std::vector<int>& objs, size_t start, size_t end) {
        int idx = out->size();
        out->box.resize(out->box.size() + 6);
        out->skip.push_back(0);
        out->prim0.push_back(-1);
        out->prim1.push_back(-1);
        out->right.push_back(-1);
        double bb[6];
        std::memcpy(bb, &pbox[(size_t)objs[start] * 6], sizeof bb);
        for (size_t i = start + 1; i < end; ++i) {  // BoundingBox::merge, src/BoundingBox.cpp:10-22
            const double* o = &pbox[(size_t)objs[i] * 6];
            for (int k = 0; k < 3; ++k) {
                bb[k] = std::min(bb[k], o[k]);
                bb[3 + k] = std::max(bb[3 + k], o[3 + k]);
            }
        }
        std::memcpy(&out->box[(size_t)idx * 6], bb, sizeof bb);
        size_t span = end - start;
        if (span == 1) {
            out->prim0[idx] = objs[start];
        } else if (span == 2) {
            out->prim0[idx] = objs[start];
            out->prim1[idx] = objs[start + 1];
        } else {
            double sx = bb[3] - bb[0], sy = bb[4] - bb[1], sz = bb[5] - bb[2];
            int axis = 0;
            if (sy > sx) axis = 1;
            if (sz > (axis == 0 ? sx : sy)) axis = 2;
            std::vector<int> sorted(objs.begin() + start, objs.begin() + end);
            const double* c = pcenter.data();
            std::sort(sorted.begin(), sorted.end(),
                      [c, axis](int a, int b) { return c[(size_t)a * 3 + axis] < c[(size_t)b * 3 + axis]; });
            size_t mid = span / 2;
            build(sorted, 0, mid);
            int r = build(sorted, mid, sorted.size());
            out->right[idx] = r;
        }
        out->skip[idx] = out->size();
        return idx;
    }
};
}  // namespace

void build_flat_bvh(const rt_scene_desc* s, FlatBVH& out) {
    out = FlatBVH();
    if (s->n_prims <= 0) return;
    Builder b;
    b.s = s;
    b.out = &out;
    b.pbox.resize((size_t)s->n_prims * 6);
    b.pcenter.resize((size_t)s->n_prims * 3);
    for (int i = 0; i < s->n_prims; ++i) {
        double* bx = &b.pbox[(size_t)i * 6];
        prim_bounds(s, i, bx);
        for (int k = 0; k < 3; ++k) b.pcenter[(size_t)i * 3 + k] = (bx[k] + bx[3 + k]) * 0.5;
    }
    std::vector<int> ids(s->n_prims);
    for (int i = 0; i < s->n_prims; ++i) ids[i] = i;
    b.build(ids, 0, ids.size());
}

}  // namespace rt

// ---------------------------------------------------------------------------------------------
struct rt_scene_file {
    std::vector<int32_t> prim_type, prim_material;
    std::vector<double> prim_geom;
    std::vector<rt_material> materials;
    std::vector<std::vector<uint8_t>> texture_bytes;
    std::vector<rt_texture> textures;
    std::vector<rt_light> lights;
    rt_scene_desc desc;
    rt_camera camera;
    rt_params params;
    // parseCamera inputs, kept so that rt_scene_file_resize can rebuild the camera
    double cam_pos[3], cam_look[3], cam_up[3], cam_fov, cam_aperture, cam_focus;
};

namespace {
using mini_json::Value;

void vec3_from(const Value& a, double out[3]) {
    out[0] = a[(size_t)0].as_double();
    out[1] = a[(size_t)1].as_double();
    out[2] = a[(size_t)2].as_double();
}

// Material::loadTexture (src/Material.cpp:17-67).  Returns false when the reference would
// clear hasTexture.
bool load_p6(const std::string& path, int& w, int& h, std::vector<uint8_t>& bytes) {
    if (path.empty()) {
        std::cerr << "Error: No texture path specified" << std::endl;
        return true;  // the reference returns without clearing hasTexture; texture stays empty
    }
    std::ifstream file(path, std::ios::binary);
    if (!file.is_open()) {
        std::cerr << "Error: Could not open texture file " << path << std::endl;
        return false;
    }
    std::string line;
    std::getline(file, line);
    if (line != "P6") {
        std::cerr << "Error: Unsupported PPM format (must be P6)" << std::endl;
        return false;
    }
    while (std::getline(file, line)) {
        if (line[0] != '#') break;
    }
    std::istringstream iss(line);
    w = 0;
    h = 0;
    iss >> w >> h;
    int maxv = 0;
    file >> maxv;
    file.get();
    long long n = (long long)w * h;
    if (n < 0) n = 0;
    bytes.assign((size_t)n * 3, 0);
    file.read(reinterpret_cast<char*>(bytes.data()), (std::streamsize)bytes.size());
    return true;
}

struct Ingest {
    rt_scene_file* f;
    std::map<std::string, int> texture_cache;

    // parseMaterial (RayTracer.cpp:1004-1035) + Material ctor (include/Material.h:30-47)
    int parse_material(const Value& m) {
        rt_material out;
        std::memset(&out, 0, sizeof out);
        out.ks = m.value("ks", 0.0);
        out.kd = m.value("kd", 0.0);
        out.specular_exponent = m.value("specularexponent", 1);
        vec3_from(m["diffusecolor"], out.diffuse);
        vec3_from(m["specularcolor"], out.specular);
        out.is_reflective = m.value("isreflective", false) ? 1 : 0;
        out.reflectivity = m.value("reflectivity", 0.0);
        out.is_refractive = m.value("isrefractive", false) ? 1 : 0;
        out.refractive_index = m.value("refractiveindex", 1.0);
        out.texture = -1;
        if (m.contains("texturepath")) {
            std::string path = m["texturepath"].as_string();
            std::cout << "Loading texture from file: " << path << std::endl;
            auto it = texture_cache.find(path);
            if (it != texture_cache.end()) {
                out.texture = it->second;
            } else {
                int w = 0, h = 0;
                std::vector<uint8_t> bytes;
                if (load_p6(path, w, h, bytes)) {
                    f->texture_bytes.push_back(std::move(bytes));
                    rt_texture t;
                    t.width = w;
                    t.height = h;
                    t.rgb = nullptr;  // fixed up in finish()
                    f->textures.push_back(t);
                    out.texture = (int)f->textures.size() - 1;
                }
                texture_cache[path] = out.texture;
            }
            std::cout << "Texture loaded successfully." << std::endl;
        }
        f->materials.push_back(out);
        return (int)f->materials.size() - 1;
    }

    // Material::Material() (src/Material.cpp:9-13) for shapes without a "material" key
    int default_material() {
        rt_material out;
        std::memset(&out, 0, sizeof out);
        out.refractive_index = 1.0;
        out.texture = -1;
        f->materials.push_back(out);
        return (int)f->materials.size() - 1;
    }

    // parseShapes (RayTracer.cpp:931-999)
    void parse_shapes(const Value& shapes) {
        if (!shapes.is_array()) return;
        for (size_t k = 0; k < shapes.size(); ++k) {
            const Value& sh = shapes[k];
            std::string type = sh["type"].as_string();
            int mat = sh.contains("material") ? parse_material(sh["material"]) : default_material();
            double g[RT_GEOM_STRIDE] = {0};
            int t;
            if (type == "sphere") {
                t = RT_SPHERE;
                vec3_from(sh["center"], g);
                g[3] = sh["radius"].as_double();
            } else if (type == "triangle") {
                t = RT_TRIANGLE;
                vec3_from(sh["v0"], g);
                vec3_from(sh["v1"], g + 3);
                vec3_from(sh["v2"], g + 6);
            } else if (type == "cylinder") {
                t = RT_CYLINDER;
                double c[3], ax[3];
                vec3_from(sh["center"], c);
                vec3_from(sh["axis"], ax);
                double radius = sh["radius"].as_double();
                double height = sh["height"].as_double();
                height *= 2.0;  // RayTracer.cpp:987
                rt::H3 base = rt::sub(rt::h3(c), rt::div(rt::mul(rt::h3(ax), height), 2.0));  // :988
                rt::H3 axis = rt::normalize(rt::h3(ax));                                      // :989
                rt::put(g, base);
                rt::put(g + 3, axis);
                g[6] = radius;
                g[7] = height;
            } else {
                std::cerr << "Error: Unsupported shape type '" << type << "'" << std::endl;
                continue;
            }
            f->prim_type.push_back(t);
            f->prim_material.push_back(mat);
            f->prim_geom.insert(f->prim_geom.end(), g, g + RT_GEOM_STRIDE);
        }
    }

    // parseLights (RayTracer.cpp:874-925)
    void parse_lights(const Value& lights) {
        if (!lights.is_array()) return;
        for (size_t k = 0; k < lights.size(); ++k) {
            const Value& lj = lights[k];
            std::string type = lj["type"].as_string();
            rt_light L;
            std::memset(&L, 0, sizeof L);
            if (type == "pointlight") {
                L.type = RT_LIGHT_POINT;
                vec3_from(lj["position"], L.position);
                vec3_from(lj["intensity"], L.intensity);
            } else if (type == "arealight") {
                L.type = RT_LIGHT_AREA;
                vec3_from(lj["position"], L.position);
                vec3_from(lj["normal"], L.normal);
                vec3_from(lj["u"], L.u);
                vec3_from(lj["v"], L.v);
                L.width = lj["width"].as_double();
                L.height = lj["height"].as_double();
                vec3_from(lj["intensity"], L.intensity);
            } else {
                std::cerr << "Error: Unsupported light type '" << type << "'" << std::endl;
                continue;
            }
            f->lights.push_back(L);
        }
    }

    void finish() {
        for (size_t i = 0; i < f->textures.size(); ++i) f->textures[i].rgb = f->texture_bytes[i].data();
        rt_scene_desc& d = f->desc;
        d.n_prims = (int32_t)f->prim_type.size();
        d.n_materials = (int32_t)f->materials.size();
        d.n_textures = (int32_t)f->textures.size();
        d.n_lights = (int32_t)f->lights.size();
        d.prim_type = f->prim_type.data();
        d.prim_material = f->prim_material.data();
        d.prim_geom = f->prim_geom.data();
        d.materials = f->materials.data();
        d.textures = f->textures.data();
        d.lights = f->lights.data();
    }
};
}  // namespace

extern "C" {

const char* rt_last_error(void) { return rt::g_last_error.c_str(); }
int rt_api_version(void) { return RT_API_VERSION; }

int rt_camera_look_at(rt_camera* out, const double position[3], const double look_at[3], const double up[3],
                      double fov_deg, double aspect, double aperture, double focus_dist) {
    if (!out || !position || !look_at || !up) return rt::set_error(RT_ERR_INVALID, "rt_camera_look_at: null argument");
    using namespace rt;
    H3 pos = h3(position);
    double lens_radius = aperture / 2.0;                     // src/Camera.cpp:15
    H3 w = normalize(sub(pos, h3(look_at)));                 // :18
    H3 u = normalize(cross(h3(up), w));                      // :19
    H3 v = cross(w, u);                                      // :20
    double theta = fov_deg * M_PI / 180.0;                   // :23
    double half_h = tan(theta / 2.0);                        // :24
    double half_w = aspect * half_h;                         // :25
    // :28  position - u*halfWidth*focusDist - v*halfHeight*focusDist - w*focusDist
    H3 ll = sub(sub(sub(pos, mul(mul(u, half_w), focus_dist)), mul(mul(v, half_h), focus_dist)), mul(w, focus_dist));
    H3 hor = mul(mul(mul(u, 2.0), half_w), focus_dist);      // :29
    H3 ver = mul(mul(mul(v, 2.0), half_h), focus_dist);      // :30
    put(out->position, pos);
    put(out->lower_left, ll);
    put(out->horizontal, hor);
    put(out->vertical, ver);
    put(out->u, u);
    put(out->v, v);
    out->lens_radius = lens_radius;
    return RT_OK;
}

int rt_scene_file_load(const char* json_path, const char* tonemap_name, rt_scene_file** out) {
    if (!json_path || !out) return rt::set_error(RT_ERR_INVALID, "rt_scene_file_load: null argument");
    *out = nullptr;
    std::ifstream in(json_path, std::ios::binary);
    if (!in.is_open()) {
        std::cerr << "Error: Could not open JSON file " << json_path << std::endl;  // RayTracer.cpp:58
        return rt::set_error(RT_ERR_IO, "could not open JSON file %s", json_path);
    }
    std::stringstream ss;
    ss << in.rdbuf();
    std::unique_ptr<rt_scene_file> f(new rt_scene_file());
    std::memset(&f->desc, 0, sizeof f->desc);
    std::memset(&f->camera, 0, sizeof f->camera);
    std::memset(&f->params, 0, sizeof f->params);
    try {
        Value root = mini_json::parse(ss.str());
        rt_params& P = f->params;
        // parseSceneSettings (RayTracer.cpp:817-832)
        P.max_depth = root.value("nbounces", 5);
        std::string mode = root.value("rendermode", "phong");
        vec3_from(root["scene"]["backgroundcolor"], f->desc.background);
        // parseCamera (RayTracer.cpp:837-868)
        const Value& cj = root["camera"];
        P.width = cj["width"].as_int();
        P.height = cj["height"].as_int();
        double aspect = static_cast<double>(P.width) / P.height;
        double pos[3], look[3], up[3];
        vec3_from(cj["position"], pos);
        vec3_from(cj["lookAt"], look);
        vec3_from(cj["upVector"], up);
        double fov = cj["fov"].as_double();
        P.exposure = cj.value("exposure", 1.0);
        double aperture = cj.value("aperture", 0.0);
        double focus = cj.value("focusDistance", rt::length(rt::sub(rt::h3(look), rt::h3(pos))));
        rt_camera_look_at(&f->camera, pos, look, up, fov, aspect, aperture, focus);
        std::memcpy(f->cam_pos, pos, sizeof pos);
        std::memcpy(f->cam_look, look, sizeof look);
        std::memcpy(f->cam_up, up, sizeof up);
        f->cam_fov = fov;
        f->cam_aperture = aperture;
        f->cam_focus = focus;

        Ingest ing;
        ing.f = f.get();
        ing.parse_lights(root["scene"]["lightsources"]);
        ing.parse_shapes(root["scene"]["shapes"]);
        ing.finish();
        f->desc.use_bvh = root.value("bvh", true) ? 1 : 0;  // RayTracer.cpp:95

        if (mode == "phong") P.mode = RT_MODE_PHONG;
        else if (mode == "binary") P.mode = RT_MODE_BINARY;
        else if (mode == "pathtrace") P.mode = RT_MODE_PATHTRACE;
        else {
            std::cerr << "Error: Unsupported rendermode '" << mode << "'. Defaulting to 'phong'." << std::endl;
            P.mode = RT_MODE_PHONG;
        }
        P.tonemap = RT_TONEMAP_NONE;
        if (tonemap_name) {  // RayTracer.cpp:116-127
            std::string tm = tonemap_name;
            if (tm == "reinhard") P.tonemap = RT_TONEMAP_REINHARD;
            else if (tm == "ward") P.tonemap = RT_TONEMAP_WARD;
            else if (tm == "uncharted2") P.tonemap = RT_TONEMAP_UNCHARTED2;
            else std::cerr << "Error: Unsupported tonemapping '" << tm << "'. Defaulting to 'none'." << std::endl;
        }
        // The swap: setLightSample(json "pixelsample"), setPixelSample(json "lightsample")
        // (RayTracer.cpp:134-139).  In the other modes the members stay unset; 16 / 4 are
        // recorded here and never used.
        int nspp = root.value("pixelsample", 16);
        int nspal = root.value("lightsample", 4);
        P.light_samples = nspp;
        P.pixel_samples = nspal;
        P.shadow_bias = 1e-4;  // include/RayTracer.h:41
        P.seed = 0;
        P.band_rows = 8;
        P.band_rank = 0;
        P.band_world = 1;
        P.sample_begin = 0;
        P.sample_end = 0;
    } catch (const std::exception& e) {
        return rt::set_error(RT_ERR_IO, "%s: %s", json_path, e.what());
    }
    *out = f.release();
    return RT_OK;
}

void rt_scene_file_free(rt_scene_file* f) { delete f; }

int rt_scene_file_resize(rt_scene_file* f, int32_t width, int32_t height) {
    if (!f || width < 2 || height < 2) return rt::set_error(RT_ERR_INVALID, "rt_scene_file_resize: bad argument");
    f->params.width = width;
    f->params.height = height;
    double aspect = static_cast<double>(width) / height;  // RayTracer.cpp:840
    return rt_camera_look_at(&f->camera, f->cam_pos, f->cam_look, f->cam_up, f->cam_fov, aspect, f->cam_aperture,
                             f->cam_focus);
}
const rt_scene_desc* rt_scene_file_desc(const rt_scene_file* f) { return f ? &f->desc : nullptr; }
const rt_camera* rt_scene_file_camera(const rt_scene_file* f) { return f ? &f->camera : nullptr; }
const rt_params* rt_scene_file_params(const rt_scene_file* f) { return f ? &f->params : nullptr; }

int rt_build_bvh(const rt_scene_desc* scene, int32_t cap, double* node_box, int32_t* node_left, int32_t* node_right) {
    if (!scene) return rt::set_error(RT_ERR_INVALID, "rt_build_bvh: null scene");
    rt::FlatBVH bvh;
    rt::build_flat_bvh(scene, bvh);
    int n = bvh.size();
    if (node_box || node_left || node_right) {
        if (cap < n) return rt::set_error(RT_ERR_LIMIT, "rt_build_bvh: capacity %d < %d nodes", cap, n);
        for (int i = 0; i < n; ++i) {
            if (node_box) std::memcpy(node_box + (size_t)i * 6, &bvh.box[(size_t)i * 6], 6 * sizeof(double));
            int32_t l, r;
            if (bvh.right[i] >= 0) {
                l = i + 1;
                r = bvh.right[i];
            } else if (bvh.prim1[i] >= 0) {
                l = ~bvh.prim0[i];
                r = ~bvh.prim1[i];
            } else {
                l = ~bvh.prim0[i];
                r = INT32_MIN;
            }
            if (node_left) node_left[i] = l;
            if (node_right) node_right[i] = r;
        }
    }
    return n;
}

}  // extern "C"

namespace {
// P3 body shared by the two writers: "r g b\n" per pixel, rows from j = H-1 down (RayTracer.cpp:352-367).  A
// table of the decimal strings keeps the 4K case (25 M numbers) quick.  value(row, k) -> the integer to print.
template <typename Row, typename Value>
int write_p3(const char* path, const Row* data, int32_t width, int32_t height, Value value) {
    FILE* fp = std::fopen(path, "wb");
    if (!fp) return rt::set_error(RT_ERR_IO, "could not open %s for writing", path);
    std::fprintf(fp, "P3\n%d %d\n255\n", width, height);
    char digits[256][4];
    int dlen[256];
    for (int v = 0; v < 256; ++v) dlen[v] = std::snprintf(digits[v], 4, "%d", v);
    // a component prints at most 11 characters ("-2147483648": out-of-range or NaN input) plus its separator
    std::vector<char> line(std::max<size_t>((size_t)width * 12, 64) + 64);
    for (int j = height - 1; j >= 0; --j) {
        char* p = line.data();
        const Row* row = data + (size_t)j * width * 3;
        for (int i = 0; i < width; ++i) {
            if ((size_t)(p - line.data()) + 40 > line.size()) {  // room for one whole pixel, whatever its values
                std::fwrite(line.data(), 1, (size_t)(p - line.data()), fp);
                p = line.data();
            }
            for (int c = 0; c < 3; ++c) {
                int v = value(row[i * 3 + c]);
                if (v >= 0 && v < 256) {
                    std::memcpy(p, digits[v], (size_t)dlen[v]);
                    p += dlen[v];
                } else {
                    p += std::sprintf(p, "%d", v);  // out-of-range input; the reference prints it too
                }
                *p++ = (c == 2) ? '\n' : ' ';
            }
        }
        std::fwrite(line.data(), 1, (size_t)(p - line.data()), fp);
    }
    if (std::fclose(fp) != 0) return rt::set_error(RT_ERR_IO, "write to %s failed", path);
    return RT_OK;
}
}  // namespace

extern "C" {

int rt_write_ppm(const char* path, const double* rgb, int32_t width, int32_t height) {
    if (!path || !rgb || width <= 0 || height <= 0) return rt::set_error(RT_ERR_INVALID, "rt_write_ppm: bad argument");
    // int(255.999 * c) as in RayTracer.cpp:359-362
    return write_p3(path, rgb, width, height, [](double c) { return static_cast<int>(255.999 * c); });
}

int rt_write_ppm_u8(const char* path, const uint8_t* rgb8, int32_t width, int32_t height) {
    if (!path || !rgb8 || width <= 0 || height <= 0) return rt::set_error(RT_ERR_INVALID, "rt_write_ppm_u8: bad argument");
    return write_p3(path, rgb8, width, height, [](uint8_t v) { return (int)v; });
}

}  // extern "C"
