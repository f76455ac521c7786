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
#define RT_TU_MAIN 1
#include "rt_device.hpp"

namespace {

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

// rows of (I, N, eta_t, eta_i) -> (refract, transmits, reflect, fresnel, fresnelReflectance); rows of colours -> the
// four tone maps: the device functions of RayTracer.cpp:430-504 and :161-202 on chosen inputs
__global__ void debug_optics_kernel(int op, long long n, const double* in, double* out, SceneView sv) {
    long long k = (long long)blockIdx.x * blockDim.x + threadIdx.x;
    if (k >= n) return;
    if (op >= 2) {  // getUV of a primitive of the uploaded scene and its material's getTextureColor (op 3: at a given (u, v))
        const double* r = in + 4 * k;
        const int ref = ((const int*)(sv.image + sv.off_ref))[(int)r[0]];
        const DMaterial& m = material_of(sv.image, sv, ref);
        double u = r[1], v = r[2];
        if (op == 2) prim_uv(sv.image, sv, ref, ld3(r + 1), u, v);
        const D3 c = texture_color(sv, m, u, v);
        double* o = out + 5 * k;
        o[0] = u; o[1] = v; o[2] = c.x; o[3] = c.y; o[4] = c.z;
    } else if (op == 0) {
        const double* r = in + 8 * k;
        double* o = out + 9 * k;
        D3 I = ld3(r), N = ld3(r + 3);
        D3 t = refract_v(I, N, r[6], r[7]);
        D3 tn = normalize(t);
        D3 rf = reflect_v(I, N);
        o[0] = t.x; o[1] = t.y; o[2] = t.z;
        o[3] = length(tn) > 0.0 ? 1.0 : 0.0;
        o[4] = rf.x; o[5] = rf.y; o[6] = rf.z;
        o[7] = fresnel_f(I, N, r[6], r[7]);
        o[8] = schlick(fabs(dot(I, N)), r[6]);
    } else {
        D3 c = ld3(in + 3 * k);
        for (int tm = 0; tm < 4; ++tm) {
            D3 v = tone_map(c, tm);
            out[12 * k + 3 * tm] = v.x;
            out[12 * k + 3 * tm + 1] = v.y;
            out[12 * k + 3 * tm + 2] = v.z;
        }
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
        case 11: r = div3(mk(x, 0.75 * x, -1.5 * x), y).x; break;  // D3 / double through one shared reciprocal when exact
        case 12: r = div3(mk(0.75 * x, -1.5 * x, x), y).z; break;
        case 13: r = sqrt_windowed(x); break;  // the library's sqrt without its rescaling wrapper when that is the identity
        case 14: {  // pq_kernel's exact accumulator: x + y + 3 x - y as 128-bit fixed point (each term truncated to 2^-80), rounded once
            unsigned long long lo = 0, hi = 0, l, h;
            const double t[4] = {x, y, 3.0 * x, -y};
            for (int k = 0; k < 4; ++k) {
                pq_fixed(t[k], l, h);
                const unsigned long long old = lo;
                lo += l;
                hi += h + (lo < old ? 1ull : 0ull);
            }
            r = pq_acc_value(lo, hi);
            break;
        }
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

// Workspace of rt_render_device_u8_static (static_host.inc): device memory that exists from the moment the code
// object is loaded on a GPU, so that the entry point needs no allocation.  Every GPU holds its own copy at its own
// address.  In host code this name is only a handle for hipGetSymbolAddress -- never a device address.
static __device__ __align__(256) unsigned char scratch[256ull << 20];

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
    bool pt_exact = false;       // RT_PT_EXACT=1: pt_kernel evaluates an area-light sample in the reference's operation order (no regrouping)
    int pt_classes = 2;          // RT_PT_CLASSES=0: pt_kernel treats every hit as general; 1: dark hits only; 2: dark and clear-shaft hits
    bool pt_queues = true;       // RT_PT_QUEUES=0 (or RT_PIPELINE=block): path trace with pt_kernel (levels per workgroup) where pq_kernel (queues per wave) applies
    double* d_pq_stack = nullptr;    // pq_kernel: per-wave ray stacks
    double* d_pq_hitq = nullptr;     //            per-wave hit queues
    unsigned* d_pq_job = nullptr;    //            job counter
    size_t pq_stack_bytes = 0, pq_hitq_bytes = 0, pq_job_bytes = 0;
    unsigned long long pq_order_gen = 0;  // generation whose job order (most expensive first) d_pq_job holds
    unsigned pq_order_jobs = 0;
    int pq_order_batch = 0;
    unsigned pq_order_min_piece = 0, pq_order_job_roots = 0;  // the order is that of jobs of this size, cut down to pieces of that size
    bool last_was_pq = false;
    int last_launches = 1;  // launches of the frame's main kernel in the last render (rt_stats.kernel_launches)
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
    DLight light0;  // host copy of the first light: pt_kernel takes it as a kernel argument
    void* scratch_ptr = nullptr;                  // this device's copy of `scratch` (hipGetSymbolAddress at rt_create)
    unsigned long long* last_counters = nullptr;  // the tallies the last render wrote (d_counters, or inside scratch)
    // wavefront node store
    double* d_wf_nodes = nullptr;
    int* d_wf_alive = nullptr;
    unsigned char* d_wf_shadow = nullptr;
    void* d_wf_state = nullptr;
    // Overflow flag of the per-workgroup pipelines (fused / block / path / grid): ONE word of pinned, device-mapped
    // host memory.  A kernel whose arena was too small sets it; the host reads it without any HIP call, so every
    // entry point can check it: the blocking ones after their copy (and render the frame again), the
    // asynchronous ones at the next call on the context (the flag stays set until the host has seen it).
    unsigned* h_overflow = nullptr;  // host address
    unsigned* d_overflow = nullptr;  // the same word as the device sees it
    long long debug_cap = 0;         // RT_WF_DEBUG_CAP: initial node capacity of an unsized setup (tests force growth)
    double debug_shrink = 0.0;       // RT_WF_DEBUG_SHRINK: factor applied ONCE to a measured capacity (tests force a later overflow)
    long long wf_cap = 0;
    size_t wf_shadow_bytes = 0;
    unsigned long long wf_calibrated_gen = 0;
    long long wf_needed_cap = 0;
    int wf_batch = 1, wf_levels_used = 0;
    bool last_was_wavefront = false;
    bool overflow_reported = false;  // the last RT_ERR_LIMIT of launch_* reported an EARLIER frame's overflow (this one was not rendered)
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
#include "pathqueue_host.inc"
#include "static_host.inc"
#include "gridwave_host.inc"

int launch_render(rt_context* c, hipStream_t stream, double* rgb_dev, int* hit_dev, bool accumulate_only,
                  unsigned char* rgb8_dev = nullptr, bool static_workspace = false) {
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
    A.no_jitter = (p.debug_flags & RT_DEBUG_NO_JITTER) ? 1 : 0;
    A.rgb = rgb_dev;
    A.rgb8 = rgb8_dev;
    A.hit_id = hit_dev;
    A.ticket = c->d_ticket;
    A.counters = c->d_counters;

    c->last_counters = c->d_counters;
    c->last_launches = 1;
    if (static_workspace) return launch_fused_static(c, stream, A);
    c->last_was_wavefront = c->wavefront;
    if (c->wavefront) {
        c->scene_in_lds = c->sv.lds_bytes <= kMaxLdsImage ? 1 : 0;
        if (c->level_kernels) return launch_wavefront(c, stream, A, accumulate_only);
        if (c->gridwave && p.mode != RT_MODE_PATHTRACE && p.max_depth > 0) return launch_gridwave(c, stream, A);
        c->last_was_pq = pq_applies(c, A);
        if (c->last_was_pq) return launch_pathqueue(c, stream, A, accumulate_only);
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

// After the device has finished the last frame (the caller has synchronized): did it flag an arena that was too
// small?  Clears the flag and forgets the measured capacities, so that the next render sizes the store again.
int frame_overflowed(rt_context* c, bool* out) {
    *out = false;
    if (!c->last_was_wavefront || c->params.max_depth <= 0) return RT_OK;
    unsigned overflow = 0;
    if (c->level_kernels) {
        if (!c->d_wf_state) return RT_OK;
        WfState st;
        HIP_TRY(hipMemcpy(&st, c->d_wf_state, sizeof st, hipMemcpyDeviceToHost));
        overflow = st.overflow;
    } else {
        overflow = *(volatile unsigned*)c->h_overflow;
        *(volatile unsigned*)c->h_overflow = 0;
    }
    if (overflow) {
        c->wf_calibrated_gen = 0;
        if (c->last_was_pq) c->pt_queues = false;  // (a ray stack ran out, which pq_kernel's bound excludes: pt_kernel from now on)
        *out = true;
    }
    return RT_OK;
}

// Blocking entry points: launch, wait, and render the frame again (sizing the node store anew) if a capacity
// measured on an earlier frame turned out too small for this one.
template <typename Launch>
int render_until_it_fits(rt_context* c, Launch launch) {
    for (int attempt = 0;; ++attempt) {
        int rc = launch();
        if (rc == RT_ERR_LIMIT && c->overflow_reported && attempt < 4) {  // an earlier asynchronous frame's flag: sized anew, go on
            c->overflow_reported = false;
            continue;
        }
        if (rc != RT_OK) return rc;
        HIP_TRY(hipDeviceSynchronize());
        bool again = false;
        if ((rc = frame_overflowed(c, &again)) != RT_OK) return rc;
        if (!again) return RT_OK;
        if (attempt >= 4) return rt::set_error(RT_ERR_LIMIT, "render: the node store overflowed %d times in a row", attempt + 1);
    }
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
    c->pt_queues = !(pipe && std::strcmp(pipe, "block") == 0);
    if (const char* e = std::getenv("RT_PT_QUEUES")) c->pt_queues = std::atoi(e) != 0;
    if (const char* e = std::getenv("RT_PT_GENERIC")) c->pt_generic = std::atoi(e) != 0;
    if (const char* e = std::getenv("RT_PT_EXACT")) c->pt_exact = std::atoi(e) != 0;
    if (const char* e = std::getenv("RT_PT_CLASSES")) c->pt_classes = std::min(2, std::max(0, std::atoi(e)));
    if (const char* e = std::getenv("RT_WF_DEBUG_CAP")) c->debug_cap = std::max(0ll, std::atoll(e));
    if (const char* e = std::getenv("RT_WF_DEBUG_SHRINK")) c->debug_shrink = std::atof(e);
    c->n_cus = prop.multiProcessorCount;
    if (const char* e = std::getenv("RT_WORKGROUPS")) c->max_workgroups = std::max(0, std::atoi(e));
    std::memset(&c->sv, 0, sizeof c->sv);
    hipError_t a = hipMalloc((void**)&c->d_ticket, sizeof(unsigned int));
    hipError_t b = hipMalloc((void**)&c->d_counters, (size_t)kCounterSlots * kCounterRow * sizeof(unsigned long long));
    hipError_t s0 = hipEventCreate(&c->ev_start);
    hipError_t s1 = hipEventCreate(&c->ev_stop);
    hipError_t h0 = hipHostMalloc((void**)&c->h_overflow, sizeof(unsigned), hipHostMallocMapped);
    hipError_t h1 = h0 == hipSuccess ? hipHostGetDevicePointer((void**)&c->d_overflow, c->h_overflow, 0) : h0;
    if (h1 == hipSuccess) *c->h_overflow = 0;
    // this device's copy of the static workspace (the device is current: hipSetDevice above)
    hipError_t sy = hipGetSymbolAddress(&c->scratch_ptr, HIP_SYMBOL(scratch));
    if (sy != hipSuccess) {
        rt_destroy(c);
        return rt::set_error(RT_ERR_DEVICE, "rt_create: hipGetSymbolAddress(scratch) on device %d -> %s", device, hipGetErrorString(sy));
    }
    if (a != hipSuccess || b != hipSuccess || s0 != hipSuccess || s1 != hipSuccess || h1 != hipSuccess) {
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
    (void)hipFree(c->d_pq_stack);
    (void)hipFree(c->d_pq_hitq);
    (void)hipFree(c->d_pq_job);
    (void)hipFree(c->d_gw_lvl);
    (void)hipFree(c->d_gw_barrier);
    (void)hipFree(c->d_k1_count);
    (void)hipFree(c->d_gw_timing);
    (void)hipFree(c->d_ticket);
    (void)hipFree(c->d_counters);
    if (c->h_overflow) (void)hipHostFree(c->h_overflow);
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
    // (sizes in 64 bits: ~370 B per primitive wrap a 32-bit sum near 5 M primitives, and the offsets below are ints)
    long long total_bytes = 0;
    {
        const long long np = s->n_prims;
        total_bytes += fb.ok ? (long long)fb.n_nodes * kFNodeBytes + np * (kLeafBytes + 32) : 0;
        total_bytes += ((long long)sph.size() + (long long)tri.size() + (long long)cyl.size()) * 8;
        total_bytes += np * 4 * 3 + (long long)s->n_materials * (long long)sizeof(DMaterial);
        total_bytes += (long long)sv.n_nodes * kNodeBytes + 16 * 16;
        if (total_bytes >= 0x7FFFFFFFll)
            return rt::set_error(RT_ERR_LIMIT, "rt_upload_scene: scene image of %lld bytes exceeds the 2 GB the 32-bit offsets address", total_bytes);
    }
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
    if (const char* e = std::getenv("RT_STAGE_MATERIALS")) if (std::atoi(e) == 0) sv.mat_staged = 0;  // A/B: records from the global copy
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
    sv.off_leafbox = off;
    if (fb.ok) off = align16(off + s->n_prims * 32);  // read by pt_classify only, from the global copy
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
            const float* lb = &fb.leaf_box[(size_t)k * 6];  // lo xyz, hi xyz -> the filter nodes' (lo, hi) pairs per axis
            const float pairs[8] = {lb[0], lb[3], lb[1], lb[4], lb[2], lb[5], 0.f, 0.f};
            std::memcpy(image.data() + sv.off_leafbox + (size_t)k * 32, pairs, 32);
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
        const double norm = (m.specular_exponent + 2.0) / (2.0 * M_PI);
        for (int k = 0; k < 3; ++k) d.spec_brdf[k] = m.specular[k] * m.ks * norm;
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
    std::memset(&c->light0, 0, sizeof c->light0);
    if (!lights.empty()) c->light0 = lights[0];
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

int rt_render_device_u8_static(rt_context* c, void* stream, uint8_t* rgb8_dev, double* rgb_dev, int32_t* hit_id_dev) {
    if (!c || !rgb8_dev) return rt::set_error(RT_ERR_INVALID, "rt_render_device_u8_static: null argument");
    if (!c->have_params) return rt::set_error(RT_ERR_STATE, "rt_render_device_u8_static: params not set");
    return launch_render(c, (hipStream_t)stream, rgb_dev, hit_id_dev, false, rgb8_dev, true);
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
    int rc = render_until_it_fits(c, [&]() {
        return pt ? launch_render(c, nullptr, d_rgb, d_hit, false) : launch_render(c, nullptr, nullptr, d_hit, false, d_u8);
    });
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
    int rc = render_until_it_fits(c, [&]() { return launch_render(c, nullptr, d_rgb, d_hit, false); });
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
        HIP_TRY(hipMemcpy(rows.data(), c->last_counters ? c->last_counters : c->d_counters, rows.size() * sizeof(unsigned long long),
                          hipMemcpyDeviceToHost));
        unsigned long long cnt[8] = {0, 0, 0, 0, 0, 0, 0, 0};
        for (size_t r = 0; r < (size_t)kCounterSlots; ++r)
            for (int k = 0; k < 7; ++k) cnt[k] += rows[r * kCounterRow + k];
        out->rays = cnt[0];
        out->rays_traversed = cnt[5];
        out->kernel_launches = c->last_launches;
        out->unwalked_ops = cnt[6];
        out->box_tests = cnt[1];
        out->sphere_tests = cnt[2];
        out->triangle_tests = cnt[3];
        out->cylinder_tests = cnt[4];
        out->prim_tests = cnt[2] + cnt[3] + cnt[4];
    }
    out->upload_ms = c->upload_ms;
    out->n_bvh_nodes = c->n_bvh_nodes;
    out->scene_in_lds = c->scene_in_lds;
    // (also for a frame without timing events -- rt_render_device_u8_static under stream capture, whose replays the caller
    // has waited for: the flag is one word of host memory)
    bool overflowed = false;
    int rc = frame_overflowed(c, &overflowed);
    if (rc != RT_OK) return rc;
    if (overflowed)
        return rt::set_error(RT_ERR_LIMIT, "the last render overflowed its node store; its output is invalid, render again");
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

int rt_debug_optics(rt_context* c, int op, int64_t n, const double* in, double* out) {
    if (!c || n < 0 || !in || !out || op < 0 || op > 3) return rt::set_error(RT_ERR_INVALID, "rt_debug_optics: bad argument");
    if (op >= 2 && !c->have_scene) return rt::set_error(RT_ERR_STATE, "rt_debug_optics: op %d needs an uploaded scene", op);
    if (op >= 2)
        for (int64_t k = 0; k < n; ++k)
            if (!(in[4 * k] >= 0 && in[4 * k] < c->sv.n_prims)) return rt::set_error(RT_ERR_INVALID, "rt_debug_optics: row %lld: no such primitive", (long long)k);
    if (n == 0) return RT_OK;
    HIP_TRY(hipSetDevice(c->device));
    const int in_cols = op == 0 ? 8 : (op == 1 ? 3 : 4), out_cols = op == 0 ? 9 : (op == 1 ? 12 : 5);
    const size_t in_bytes = (size_t)n * in_cols * sizeof(double), out_bytes = (size_t)n * out_cols * sizeof(double);
    double *d_in = nullptr, *d_out = nullptr;
    int rc = RT_OK;
    hipError_t e = hipMalloc((void**)&d_in, in_bytes);
    if (e == hipSuccess) e = hipMalloc((void**)&d_out, out_bytes);
    if (e == hipSuccess) e = hipMemcpy(d_in, in, in_bytes, hipMemcpyHostToDevice);
    if (e == hipSuccess) {
        hipLaunchKernelGGL(debug_optics_kernel, dim3((unsigned)((n + 255) / 256)), dim3(256), 0, 0, op, (long long)n, d_in, d_out, c->sv);
        e = hipGetLastError();
    }
    if (e == hipSuccess) e = hipMemcpy(out, d_out, out_bytes, hipMemcpyDeviceToHost);
    if (e != hipSuccess) rc = rt::set_error(RT_ERR_DEVICE, "rt_debug_optics -> %s", hipGetErrorString(e));
    (void)hipFree(d_in); (void)hipFree(d_out);
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
