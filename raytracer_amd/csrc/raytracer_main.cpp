// raytracer_main.cpp -- the drop-in CLI:  raytracer <scene.json> <out.ppm> [reinhard|ward|uncharted2]
//
// Same invocation, banners, defaults and exit codes as the reference's main()
// (/root/reference/src/RayTracer.cpp:42-159); the render itself goes to the GPU through the
// C-ABI in include/raytracer_amd.h.  Optional environment variables (no effect on the original
// invocation): RT_DEVICE (GPU ordinal), RT_DEVICES (comma-separated GPU ordinals: the frame is split over
// them in this one process, rt_group_*), RT_WIDTH / RT_HEIGHT (override camera.width/height; the
// aspect ratio of the camera follows, as it would after editing the JSON), RT_SEED, and for path tracing
// RT_FIX_SAMPLE_SWAP=1 (undo the reference's swap at RayTracer.cpp:138-139: JSON "pixelsample" then means
// samples per pixel and "lightsample" samples per light), RT_PIXEL_SAMPLES / RT_LIGHT_SAMPLES (set
// RayTracer::pixelSamples / lightSamples directly, after the swap or its undoing).
#include <chrono>
#include <cstdint>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <iostream>
#include <string>
#include <utility>
#include <vector>

#include "raytracer_amd.h"

static int env_int(const char* name, int dflt) {
    const char* v = std::getenv(name);
    return (v && *v) ? std::atoi(v) : dflt;
}

int main(int argc, char* argv[]) {
    auto start = std::chrono::high_resolution_clock::now();
    if (!(argc == 3 || argc == 4)) {
        std::cerr << "Usage: raytracer.exe path_to_JSON output_filename.ppm <optional-tonemapping>" << std::endl;
        return 1;
    }
    std::string json_path = argv[1], out_path = argv[2];

    std::cout << "Parsing scene settings..." << std::endl;
    rt_scene_file* file = nullptr;
    int rc = rt_scene_file_load(json_path.c_str(), argc == 4 ? argv[3] : nullptr, &file);
    if (rc == RT_ERR_IO && !file) return 1;  // the loader has printed "Error: Could not open JSON file ..." (RayTracer.cpp:58)
    if (rc != RT_OK) {
        std::cerr << "Error: " << rt_last_error() << std::endl;
        return 1;
    }
    std::cout << "Scene settings parsed." << std::endl;
    std::cout << "Parsing camera settings..." << std::endl << "Camera settings parsed." << std::endl;
    std::cout << "Parsing lights..." << std::endl << "Lights parsed." << std::endl;
    std::cout << "Parsing shapes..." << std::endl << "Shapes parsed." << std::endl;

    int ow = env_int("RT_WIDTH", 0), oh = env_int("RT_HEIGHT", 0);
    if (ow > 0 && oh > 0 && rt_scene_file_resize(file, ow, oh) != RT_OK) {
        std::cerr << "Error: " << rt_last_error() << std::endl;
        return 1;
    }
    const rt_scene_desc* desc = rt_scene_file_desc(file);
    rt_params params = *rt_scene_file_params(file);
    rt_camera camera = *rt_scene_file_camera(file);
    if (desc->use_bvh) std::cout << "Building BVH..." << std::endl;
    if (params.mode == RT_MODE_PATHTRACE) {
        if (env_int("RT_FIX_SAMPLE_SWAP", 0) != 0) std::swap(params.pixel_samples, params.light_samples);
        params.pixel_samples = env_int("RT_PIXEL_SAMPLES", params.pixel_samples);
        params.light_samples = env_int("RT_LIGHT_SAMPLES", params.light_samples);
    }
    const char* seed_env = std::getenv("RT_SEED");
    if (seed_env && *seed_env) params.seed = std::strtoull(seed_env, nullptr, 10);

    // the file's payload straight from the device(s): int(255.999 * c) bytes (RayTracer.cpp:359-362)
    std::vector<uint8_t> rgb((size_t)params.width * params.height * 3);
    std::vector<int> devices;
    if (const char* list = std::getenv("RT_DEVICES")) {
        for (const char* q = list; *q;) {
            char* end = nullptr;
            long v = std::strtol(q, &end, 10);
            if (end == q) break;
            devices.push_back((int)v);
            q = (*end == ',') ? end + 1 : end;
        }
    }
    rt_context* ctx = nullptr;
    rt_group* group = nullptr;
    rt_stats st;
    auto banners = [&]() {
        if (desc->use_bvh) std::cout << "BVH built." << std::endl;
        if (params.mode == RT_MODE_PATHTRACE) {
            std::cout << "Pixel samples: " << params.pixel_samples << std::endl;
            std::cout << "Light samples: " << params.light_samples << std::endl;
        }
    };
    bool ok;
    if (devices.size() > 1) {
        ok = rt_group_create(&group, devices.data(), (int)devices.size()) == RT_OK && rt_group_upload_scene(group, desc) == RT_OK &&
             rt_group_set_camera(group, &camera) == RT_OK && rt_group_set_params(group, &params) == RT_OK;
        if (ok) banners();
        ok = ok && rt_group_render_u8(group, rgb.data(), nullptr) == RT_OK && rt_group_get_stats(group, &st) == RT_OK;
    } else {
        ok = rt_create(&ctx, devices.size() == 1 ? devices[0] : env_int("RT_DEVICE", 0)) == RT_OK && rt_upload_scene(ctx, desc) == RT_OK &&
             rt_set_camera(ctx, &camera) == RT_OK && rt_set_params(ctx, &params) == RT_OK;
        if (ok) banners();
        ok = ok && rt_render_u8(ctx, rgb.data(), nullptr) == RT_OK && rt_get_stats(ctx, &st) == RT_OK;
    }
    if (!ok) {
        std::cerr << "Error: " << rt_last_error() << std::endl;
        return 1;
    }
    std::cout << "\rRendering: 100% completed" << std::flush;
    if (rt_write_ppm_u8(out_path.c_str(), rgb.data(), params.width, params.height) != RT_OK) {
        std::cerr << "Error: " << rt_last_error() << std::endl;
        return 1;
    }
    auto end = std::chrono::high_resolution_clock::now();
    auto ms = std::chrono::duration_cast<std::chrono::milliseconds>(end - start).count();
    std::cout << "Rendering complete. Image saved to " << out_path << std::endl;
    std::cout << "Total execution time: " << ms << " milliseconds" << std::endl;
    std::cout << "GPU render: " << st.kernel_ms << " ms, " << st.rays << " rays, "
              << (st.kernel_ms > 0 ? st.rays / st.kernel_ms / 1e3 : 0.0) << " Mrays/s" << std::endl;
    rt_destroy(ctx);
    rt_group_destroy(group);
    rt_scene_file_free(file);
    return 0;
}
