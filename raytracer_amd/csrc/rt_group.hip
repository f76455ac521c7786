// rt_group.hip -- one frame over several GPUs of a node from ONE process: the rt_group_* entry points of
// include/raytracer_amd.h.  The reference's seam is a single blocking call (src/RayTracer.cpp:146-149); a group is
// what stands behind that call when the caller names more than one device.
//
// A group owns one rt_context and one stream per device; the scene is replicated (a few MB).  A frame is
// partitioned as SURVEY.md 8(e) lays out:
//   Whitted / binary   8-row bands dealt round-robin (rt_params.band_*): member k renders its bands compactly on
//                      its own GPU, and ONE strided peer copy per member (hipMemcpy2DAsync: band -> its rows of the
//                      frame, pitch = world x band) assembles the frame on the first device over xGMI.  What
//                      travels is the PPM payload (bytes) when the caller asks for bytes: an eighth of the fp64 frame.
//   path trace         strata ranges (rt_params.sample_begin / sample_end): every member accumulates the radiance
//                      sums of its strata for the whole frame on its own GPU.  The sums are then combined as a
//                      reduce-scatter by peer reads: member j owns pixels [n j / N, n (j + 1) / N), reads that slice of
//                      every member's sums, adds them in member order BEFORE the non-linear stages (/ pixelSamples,
//                      tone map, exposure, gamma, clamp: RayTracer.cpp:301-315), runs those stages (and the byte
//                      conversion) on its slice and writes the finished slice into the frame on the first device.
//                      Every GPU reads (N-1)/N of a frame and the first device receives (N-1)/N of a finished frame,
//                      whatever N is; the order of the additions, and so the frame, does not depend on N's timing.
// Every launch and copy is asynchronous on its member's stream; the first device's stream waits for the members'
// events, the caller for that stream.  The exception is the first frame of a setup, in which each context sizes its node
// store and waits for its own GPU (blockwave_host.inc): for that frame the members are driven by one host thread each,
// so the sizing frames of N GPUs run side by side instead of one after the other.  One process per GPU with RCCL
// (raytracer_amd/dist.py, bench.py) is the other way to run the same partition; both produce the same frame.
//
// Not measured on this project's hardware: the build box has one GPU, so the tests name device 0 several times
// (tests/test_gpu_group.py) -- separate contexts, streams and buffers, the same partition, copies and reduction, but no
// byte crosses xGMI.
#include <hip/hip_runtime.h>

#include <algorithm>
#include <cmath>
#include <cstring>
#include <string>
#include <thread>
#include <vector>

#include "rt_internal.hpp"

#define HIP_TRY(expr)                                                                                      \
    do {                                                                                                   \
        hipError_t e_ = (expr);                                                                            \
        if (e_ != hipSuccess) return rt::set_error(RT_ERR_DEVICE, "%s -> %s", #expr, hipGetErrorString(e_)); \
    } while (0)
#define RT_TRY(expr)                 \
    do {                             \
        int rc_ = (expr);            \
        if (rc_ != RT_OK) return rc_; \
    } while (0)

namespace {

constexpr int kMaxMembers = 64;

struct Sources {
    const double* p[kMaxMembers];
};

// dst[i] = ((src0[begin + i] + src1[begin + i]) + src2[begin + i]) + ...: one slice of the members' radiance sums, the
// sources read where they lie (peer memory for all but one of them)
__global__ void __launch_bounds__(256) group_reduce_kernel(double* dst, Sources src, int world, long long begin, long long n) {
    const long long stride = (long long)gridDim.x * blockDim.x;
    for (long long i = (long long)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += stride) {
        double s = src.p[0][begin + i];
        for (int k = 1; k < world; ++k) s = s + src.p[k][begin + i];
        dst[i] = s;
    }
}

struct Member {
    int device = 0;
    rt_context* ctx = nullptr;
    hipStream_t stream = nullptr;
    hipEvent_t done = nullptr;         // this member's part of the frame has arrived on the first device
    hipEvent_t accumulated = nullptr;  // path trace: this member's radiance sums are complete
    void* out = nullptr;      // this member's compact bands (bytes or doubles) / its radiance sums
    size_t out_bytes = 0;
    int* hit = nullptr;       // compact primary-hit ids
    size_t hit_bytes = 0;
    void* slice = nullptr;    // path trace: the reduced (then resolved) slice this member owns
    size_t slice_bytes = 0;
    unsigned char* slice_u8 = nullptr;
    size_t slice_u8_bytes = 0;
    int64_t rows = 0;
};

int ensure(void** p, size_t* have, size_t want, int device) {
    if (want <= *have) return RT_OK;
    HIP_TRY(hipSetDevice(device));
    if (*p) {
        HIP_TRY(hipDeviceSynchronize());
        (void)hipFree(*p);
        *p = nullptr;
        *have = 0;
    }
    HIP_TRY(hipMalloc(p, want));
    *have = want;
    return RT_OK;
}

}  // namespace

struct rt_group {
    std::vector<Member> m;
    rt_params params;
    bool have_params = false;
    bool fresh = true;      // no frame rendered since the setup changed: the members' contexts will size their node stores
    void* frame = nullptr;  // on the first device: the assembled frame (bytes or doubles)
    size_t frame_bytes = 0;
    unsigned char* frame_u8 = nullptr;  // path trace: the quantized frame
    size_t frame_u8_bytes = 0;
    int* frame_hit = nullptr;
    size_t frame_hit_bytes = 0;
};

namespace {

// Member k's compact bands -> their rows of the frame on the first device, on k's stream.  elem = bytes per pixel.
int scatter_bands(rt_group* g, int k, const void* src, void* frame, size_t elem) {
    const rt_params& p = g->params;
    const int world = (int)g->m.size(), br = p.band_rows;
    const size_t row = (size_t)p.width * elem, band = row * br;
    const int n_bands = (p.height + br - 1) / br;
    const int mine = k < n_bands ? (n_bands - 1 - k) / world + 1 : 0;  // bands k, k + world, ...
    if (mine == 0) return RT_OK;
    const bool ragged_last = (p.height % br) != 0 && (k + (mine - 1) * world) == n_bands - 1;
    const int full = ragged_last ? mine - 1 : mine;
    Member& me = g->m[k];
    char* dst0 = (char*)frame + (size_t)k * band;
    HIP_TRY(hipSetDevice(me.device));
    if (full > 0)
        HIP_TRY(hipMemcpy2DAsync(dst0, band * world, src, band, band, (size_t)full, hipMemcpyDeviceToDevice, me.stream));
    if (ragged_last) {
        const size_t rest = row * (size_t)(p.height % br);
        HIP_TRY(hipMemcpyAsync(dst0 + (size_t)full * band * world, (const char*)src + (size_t)full * band, rest,
                               hipMemcpyDeviceToDevice, me.stream));
    }
    return RT_OK;
}

// Member k's own work for one frame, everything on its stream: render (bands or strata), hand the bands over.
int render_member(rt_group* g, int k, bool want_u8, bool want_hit) {
    const rt_params& p = g->params;
    const bool pt = p.mode == RT_MODE_PATHTRACE;
    const size_t n_px = (size_t)p.width * p.height;
    const size_t elem = (pt || !want_u8) ? 3 * sizeof(double) : 3;
    Member& me = g->m[k];
    if (pt) {
        // member 0's sums lie in the frame buffer itself (every slice of it is read by the slice's owner before that
        // owner overwrites it, on one stream), the others' in a buffer of their own
        if (k > 0) RT_TRY(ensure(&me.out, &me.out_bytes, n_px * elem, me.device));
        RT_TRY(rt_pathtrace_accumulate_device(me.ctx, me.stream, k == 0 ? (double*)g->frame : (double*)me.out));
        HIP_TRY(hipSetDevice(me.device));
        HIP_TRY(hipEventRecord(me.accumulated, me.stream));
        return RT_OK;
    }
    const size_t mine_px = (size_t)me.rows * p.width;
    RT_TRY(ensure(&me.out, &me.out_bytes, std::max<size_t>(mine_px, 1) * elem, me.device));
    if (want_hit) {
        RT_TRY(ensure((void**)&me.hit, &me.hit_bytes, std::max<size_t>(mine_px, 1) * sizeof(int), me.device));
        HIP_TRY(hipSetDevice(me.device));
        HIP_TRY(hipMemsetAsync(me.hit, 0xFF, std::max<size_t>(mine_px, 1) * sizeof(int), me.stream));
    }
    if (mine_px > 0) {
        if (want_u8) RT_TRY(rt_render_device_u8(me.ctx, me.stream, (uint8_t*)me.out, nullptr, want_hit ? me.hit : nullptr));
        else RT_TRY(rt_render_device(me.ctx, me.stream, (double*)me.out, want_hit ? me.hit : nullptr));
    }
    RT_TRY(scatter_bands(g, k, me.out, g->frame, elem));
    if (want_hit) RT_TRY(scatter_bands(g, k, me.hit, g->frame_hit, sizeof(int)));
    HIP_TRY(hipSetDevice(me.device));
    HIP_TRY(hipEventRecord(me.done, me.stream));
    return RT_OK;
}

// Path trace, after every member has been launched: member j reduces, resolves and delivers the slice it owns.
int combine_slices(rt_group* g, bool want_u8) {
    const rt_params& p = g->params;
    const int world = (int)g->m.size();
    const long long n_px = (long long)p.width * p.height;
    Sources src;
    std::memset(&src, 0, sizeof src);
    for (int k = 0; k < world; ++k) src.p[k] = k == 0 ? (const double*)g->frame : (const double*)g->m[k].out;
    for (int j = 0; j < world; ++j) {
        Member& me = g->m[j];
        const long long lo = n_px * j / world, hi = n_px * (j + 1) / world, mine = hi - lo;
        HIP_TRY(hipSetDevice(me.device));
        if (mine > 0) {
            RT_TRY(ensure(&me.slice, &me.slice_bytes, (size_t)mine * 3 * sizeof(double), me.device));
            if (want_u8) RT_TRY(ensure((void**)&me.slice_u8, &me.slice_u8_bytes, (size_t)mine * 3, me.device));
            HIP_TRY(hipSetDevice(me.device));
            for (int k = 0; k < world; ++k)
                if (k != j) HIP_TRY(hipStreamWaitEvent(me.stream, g->m[k].accumulated, 0));
            const long long n = mine * 3;
            const unsigned grid = (unsigned)std::min<long long>((n + 255) / 256, 2048);
            hipLaunchKernelGGL(group_reduce_kernel, dim3(grid), dim3(256), 0, me.stream, (double*)me.slice, src, world, lo * 3, n);
            HIP_TRY(hipGetLastError());
            RT_TRY(rt_pathtrace_resolve_device(me.ctx, me.stream, (double*)me.slice, (int64_t)mine));
            if (want_u8) {
                RT_TRY(rt_quantize_device(me.ctx, me.stream, (const double*)me.slice, (int64_t)n, me.slice_u8));
                HIP_TRY(hipMemcpyAsync(g->frame_u8 + lo * 3, me.slice_u8, (size_t)n, hipMemcpyDeviceToDevice, me.stream));
            } else {
                HIP_TRY(hipMemcpyAsync((double*)g->frame + lo * 3, me.slice, (size_t)n * sizeof(double), hipMemcpyDeviceToDevice, me.stream));
            }
        }
        HIP_TRY(hipEventRecord(me.done, me.stream));
    }
    return RT_OK;
}

int render_frame(rt_group* g, bool want_u8, bool want_hit) {
    if (!g->have_params) return rt::set_error(RT_ERR_STATE, "rt_group: params not set");
    const rt_params& p = g->params;
    const int world = (int)g->m.size();
    const bool pt = p.mode == RT_MODE_PATHTRACE;
    const size_t n_px = (size_t)p.width * p.height;
    Member& first = g->m[0];
    const size_t elem = (pt || !want_u8) ? 3 * sizeof(double) : 3;
    RT_TRY(ensure(&g->frame, &g->frame_bytes, n_px * elem, first.device));
    if (want_hit) RT_TRY(ensure((void**)&g->frame_hit, &g->frame_hit_bytes, n_px * sizeof(int), first.device));
    if (pt && want_u8) RT_TRY(ensure((void**)&g->frame_u8, &g->frame_u8_bytes, n_px * 3, first.device));
    if (g->fresh && world > 1) {
        // the sizing frame of each context ends in a wait for its GPU: one host thread per member, so that the GPUs'
        // sizing frames overlap (error text is per thread: carried back by hand)
        std::vector<int> rcs((size_t)world, RT_OK);
        std::vector<std::string> msgs((size_t)world);
        auto run = [&](int k) {
            rcs[(size_t)k] = render_member(g, k, want_u8, want_hit);
            if (rcs[(size_t)k] != RT_OK) msgs[(size_t)k] = rt_last_error();
        };
        std::vector<std::thread> workers;
        for (int k = 1; k < world; ++k) workers.emplace_back(run, k);
        run(0);
        for (std::thread& t : workers) t.join();
        for (int k = 0; k < world; ++k)
            if (rcs[(size_t)k] != RT_OK) return rt::set_error(rcs[(size_t)k], "%s", msgs[(size_t)k].c_str());
    } else {
        for (int k = 0; k < world; ++k) RT_TRY(render_member(g, k, want_u8, want_hit));
    }
    g->fresh = false;
    if (pt) RT_TRY(combine_slices(g, want_u8));
    // the first device's stream continues once every member has delivered
    HIP_TRY(hipSetDevice(first.device));
    for (int k = 1; k < world; ++k) HIP_TRY(hipStreamWaitEvent(first.stream, g->m[k].done, 0));
    return RT_OK;
}

// Blocking: the frame in host memory.  A member whose measured node-store capacity was too small for this frame
// reports it through rt_get_stats (render_hip.hip keeps the flag); the frame is then rendered again, sized anew.
int render_to_host(rt_group* g, void* host, bool want_u8, int32_t* hit_host) {
    const size_t n_px = (size_t)g->params.width * g->params.height;
    const bool pt = g->params.mode == RT_MODE_PATHTRACE;
    for (int attempt = 0;; ++attempt) {
        int rc = render_frame(g, want_u8, hit_host != nullptr);
        bool again = false;
        if (rc == RT_OK) {
            HIP_TRY(hipSetDevice(g->m[0].device));
            HIP_TRY(hipStreamSynchronize(g->m[0].stream));
        } else {  // a launch failed part-way: let whatever the other members were given finish before anything is reused
            for (Member& me : g->m)
                if (hipSetDevice(me.device) == hipSuccess) (void)hipStreamSynchronize(me.stream);
        }
        for (Member& me : g->m) {
            rt_stats st;
            int s = rt_get_stats(me.ctx, &st);
            if (s == RT_ERR_LIMIT) again = true;
            else if (s != RT_OK && rc == RT_OK) rc = s;
        }
        if (rc == RT_ERR_LIMIT) again = true;  // an earlier asynchronous frame's overflow, reported by a launch
        else if (rc != RT_OK) return rc;
        if (!again) break;
        g->fresh = true;  // the member that overflowed sizes its store anew
        if (attempt >= 4) return rt::set_error(RT_ERR_LIMIT, "rt_group: the node store overflowed %d times in a row", attempt + 1);
    }
    HIP_TRY(hipSetDevice(g->m[0].device));
    const void* src = (pt && want_u8) ? (const void*)g->frame_u8 : g->frame;
    HIP_TRY(hipMemcpy(host, src, n_px * (want_u8 ? 3 : 3 * sizeof(double)), hipMemcpyDeviceToHost));
    if (hit_host) {
        if (pt) std::memset(hit_host, 0xFF, n_px * sizeof(int));  // path-traced frames carry no hit ids (rt_render: -1)
        else HIP_TRY(hipMemcpy(hit_host, g->frame_hit, n_px * sizeof(int), hipMemcpyDeviceToHost));
    }
    return RT_OK;
}

}  // namespace

extern "C" {

int rt_group_create(rt_group** out, const int* devices, int n_devices) {
    if (!out || !devices || n_devices < 1 || n_devices > kMaxMembers) return rt::set_error(RT_ERR_INVALID, "rt_group_create: bad argument");
    *out = nullptr;
    rt_group* g = new rt_group();
    g->m.resize((size_t)n_devices);
    int rc = RT_OK;
    for (int k = 0; k < n_devices && rc == RT_OK; ++k) {
        Member& me = g->m[k];
        me.device = devices[k];
        rc = rt_create(&me.ctx, me.device);
        if (rc != RT_OK) break;
        hipError_t e = hipSetDevice(me.device);
        if (e == hipSuccess) e = hipStreamCreateWithFlags(&me.stream, hipStreamNonBlocking);
        if (e == hipSuccess) e = hipEventCreateWithFlags(&me.done, hipEventDisableTiming);
        if (e == hipSuccess) e = hipEventCreateWithFlags(&me.accumulated, hipEventDisableTiming);
        // the member writes into the first device's memory and reads a slice of every other member's sums
        for (int j = 0; j < n_devices && e == hipSuccess && rc == RT_OK; ++j) {
            if (devices[j] == me.device) continue;
            int can = 0;
            e = hipDeviceCanAccessPeer(&can, me.device, devices[j]);
            if (e == hipSuccess && !can)
                rc = rt::set_error(RT_ERR_DEVICE, "rt_group_create: device %d cannot access device %d's memory", me.device, devices[j]);
            else if (e == hipSuccess) {
                e = hipDeviceEnablePeerAccess(devices[j], 0);
                if (e == hipErrorPeerAccessAlreadyEnabled) {
                    (void)hipGetLastError();
                    e = hipSuccess;
                }
            }
        }
        if (rc != RT_OK) break;
        if (e != hipSuccess) rc = rt::set_error(RT_ERR_DEVICE, "rt_group_create: device %d -> %s", me.device, hipGetErrorString(e));
    }
    if (rc != RT_OK) {
        rt_group_destroy(g);
        return rc;
    }
    *out = g;
    return RT_OK;
}

void rt_group_destroy(rt_group* g) {
    if (!g) return;
    for (Member& me : g->m) {
        if (!me.ctx) continue;
        (void)hipSetDevice(me.device);
        (void)hipDeviceSynchronize();
        (void)hipFree(me.out);
        (void)hipFree(me.hit);
        (void)hipFree(me.slice);
        (void)hipFree(me.slice_u8);
        if (me.accumulated) (void)hipEventDestroy(me.accumulated);
        if (me.stream) (void)hipStreamDestroy(me.stream);
        if (me.done) (void)hipEventDestroy(me.done);
    }
    if (!g->m.empty() && g->m[0].ctx) {
        (void)hipSetDevice(g->m[0].device);
        (void)hipFree(g->frame);
        (void)hipFree(g->frame_u8);
        (void)hipFree(g->frame_hit);
    }
    for (Member& me : g->m) rt_destroy(me.ctx);
    delete g;
}

int rt_group_size(const rt_group* g) { return g ? (int)g->m.size() : rt::set_error(RT_ERR_INVALID, "rt_group_size: null group"); }

int rt_group_upload_scene(rt_group* g, const rt_scene_desc* scene) {
    if (!g || !scene) return rt::set_error(RT_ERR_INVALID, "rt_group_upload_scene: null argument");
    g->fresh = true;
    for (Member& me : g->m) RT_TRY(rt_upload_scene(me.ctx, scene));
    return RT_OK;
}

int rt_group_set_camera(rt_group* g, const rt_camera* cam) {
    if (!g || !cam) return rt::set_error(RT_ERR_INVALID, "rt_group_set_camera: null argument");
    g->fresh = true;
    for (Member& me : g->m) RT_TRY(rt_set_camera(me.ctx, cam));
    return RT_OK;
}

int rt_group_set_params(rt_group* g, const rt_params* p) {
    if (!g || !p) return rt::set_error(RT_ERR_INVALID, "rt_group_set_params: null argument");
    if (p->band_world != 1 || p->band_rank != 0 || p->sample_begin != 0 || p->sample_end > 0)
        return rt::set_error(RT_ERR_INVALID, "rt_group_set_params: the group partitions the frame itself; pass whole-frame params "
                                             "(band_rank 0, band_world 1, sample_begin 0, sample_end 0)");
    const int world = (int)g->m.size();
    g->fresh = true;
    for (int k = 0; k < world; ++k) {
        rt_params q = *p;
        if (p->mode == RT_MODE_PATHTRACE) {
            const int side = p->pixel_samples > 0 ? (int)std::sqrt((double)p->pixel_samples) : 0;  // RayTracer.cpp:255
            const long long n = (long long)side * side;
            const int lo = (int)(n * k / world), hi = (int)(n * (k + 1) / world);
            q.sample_begin = hi > lo ? lo : (int)n;  // an empty share is (n, n): sample_end <= 0 would mean "all"
            q.sample_end = hi > lo ? hi : (int)n;
            if (n == 0) q.sample_begin = q.sample_end = 0;
        } else {
            q.band_rank = k;
            q.band_world = world;
        }
        RT_TRY(rt_set_params(g->m[k].ctx, &q));
        g->m[k].rows = rt_local_rows(g->m[k].ctx);
    }
    g->params = *p;
    g->have_params = true;
    return RT_OK;
}

int rt_group_render(rt_group* g, double* rgb, int32_t* hit_id) {
    if (!g || !rgb) return rt::set_error(RT_ERR_INVALID, "rt_group_render: null argument");
    if (!g->have_params) return rt::set_error(RT_ERR_STATE, "rt_group_render: params not set");
    return render_to_host(g, rgb, false, hit_id);
}

int rt_group_render_u8(rt_group* g, uint8_t* rgb8, int32_t* hit_id) {
    if (!g || !rgb8) return rt::set_error(RT_ERR_INVALID, "rt_group_render_u8: null argument");
    if (!g->have_params) return rt::set_error(RT_ERR_STATE, "rt_group_render_u8: params not set");
    return render_to_host(g, rgb8, true, hit_id);
}

int rt_group_get_stats(rt_group* g, rt_stats* out) {
    if (!g || !out) return rt::set_error(RT_ERR_INVALID, "rt_group_get_stats: null argument");
    std::memset(out, 0, sizeof *out);
    for (Member& me : g->m) {
        rt_stats st;
        RT_TRY(rt_get_stats(me.ctx, &st));
        out->rays += st.rays;
        out->rays_traversed += st.rays_traversed;
        out->unwalked_ops += st.unwalked_ops;
        out->kernel_launches = std::max(out->kernel_launches, st.kernel_launches);
        out->box_tests += st.box_tests;
        out->prim_tests += st.prim_tests;
        out->sphere_tests += st.sphere_tests;
        out->triangle_tests += st.triangle_tests;
        out->cylinder_tests += st.cylinder_tests;
        out->kernel_ms = std::max(out->kernel_ms, st.kernel_ms);  // the members render side by side
        out->upload_ms += st.upload_ms;
        out->n_bvh_nodes = st.n_bvh_nodes;
        out->scene_in_lds = st.scene_in_lds;
    }
    return RT_OK;
}

}  // extern "C"
