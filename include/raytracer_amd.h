/* raytracer_amd.h -- C-ABI of libraytracer_amd.so, the MI355X (gfx950) render loop.
 *
 * The reference (Basemism/Raytracer) has no plugin/FFI API.  Its only seam is the pair of
 * member calls made once from main():
 *
 *     rayTracer.renderPathTrace(outputFilename) / rayTracer.render(outputFilename)
 *                                   -- /root/reference/src/RayTracer.cpp:146-149
 *
 * after the setters at RayTracer.cpp:129-139.  Everything a replacement needs at that seam is
 * plain data: Scene::objects / lights / backgroundColor / bvhRoot-or-null (include/Scene.h:17-21),
 * the Camera constants (include/Camera.h:11-26) and the RayTracer scalars
 * (include/RayTracer.h:35-47).  This header states that data as POD structs and the calls as
 * extern "C" functions: plain pointers and sizes, no C++ types, no exceptions across the
 * boundary.  Every function returns 0 on success and a negative rt_status on failure;
 * rt_last_error() returns a thread-local message for the last failing call.
 *
 * Conventions
 *   - All arithmetic on the path is IEEE binary64, as in the reference (include/Vector3.h:14).
 *   - Primitive ID = index in Scene::objects = order of the JSON "shapes" array with unsupported
 *     types skipped (RayTracer.cpp:931-999, Scene.cpp:15-17).  -1 = miss.
 *   - Image buffers are row-major, row j = 0 is the BOTTOM row (the writer emits j = H-1 first,
 *     RayTracer.cpp:356), pixel (i, j) at rgb[(j*W + i)*3 .. +2].
 */
#ifndef RAYTRACER_AMD_H
#define RAYTRACER_AMD_H

#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

#define RT_API_VERSION 1

typedef enum rt_status {
    RT_OK = 0,
    RT_ERR_INVALID = -1,  /* bad argument / inconsistent sizes                     */
    RT_ERR_IO = -2,       /* file could not be opened / parsed                      */
    RT_ERR_DEVICE = -3,   /* HIP runtime error or no gfx950 device / kernel missing */
    RT_ERR_STATE = -4,    /* call order (render before upload, ...)                 */
    RT_ERR_LIMIT = -5     /* a documented size limit was exceeded                   */
} rt_status;

/* Intersectable subclasses (include/Sphere.h, Triangle.h, Cylinder.h). */
enum { RT_SPHERE = 0, RT_TRIANGLE = 1, RT_CYLINDER = 2 };
/* RayTracer::RenderMode, same numbering (include/RayTracer.h:15). */
enum { RT_MODE_PHONG = 0, RT_MODE_BINARY = 1, RT_MODE_PATHTRACE = 2 };
/* RayTracer::ToneMapping, same numbering (include/RayTracer.h:16). */
enum { RT_TONEMAP_NONE = 0, RT_TONEMAP_REINHARD = 1, RT_TONEMAP_WARD = 2, RT_TONEMAP_UNCHARTED2 = 3 };
/* Light::LightType (include/Light.h:11-14). */
enum { RT_LIGHT_POINT = 0, RT_LIGHT_AREA = 1 };

/* One geometry row per primitive, RT_GEOM_STRIDE doubles, holding the CONSTRUCTOR ARGUMENTS of
 * the reference class (the library applies what the constructors do):
 *   sphere   : center xyz, radius                              (src/Sphere.cpp:9-10)
 *   triangle : v0 xyz, v1 xyz, v2 xyz; the normal is derived as in src/Triangle.cpp:7-12
 *   cylinder : baseCenter xyz, axis xyz, radius, height -- i.e. AFTER the parse transform of
 *              RayTracer.cpp:987-989; the constructor's own axis.normalize()
 *              (src/Cylinder.cpp:11) is applied by the library.
 * Unused slots are ignored. */
#define RT_GEOM_STRIDE 12

/* Material (include/Material.h:12-26).  texture = index into rt_scene_desc.textures or -1. */
typedef struct rt_material {
    double ks, kd;
    double reflectivity, refractive_index;
    double diffuse[3];
    double specular[3];
    int32_t specular_exponent; /* int in the reference (Material.h:14) */
    int32_t is_reflective;
    int32_t is_refractive;
    int32_t texture;
} rt_material;

/* Texture as loaded by Material::loadTexture (src/Material.cpp:17-67): 8-bit RGB, row 0 first.
 * The reference stores byte/255.0 as f64; the device divides on fetch, same value. */
typedef struct rt_texture {
    int32_t width, height;
    const uint8_t* rgb; /* width*height*3 bytes */
} rt_texture;

/* PointLight / AreaLight constructor arguments (include/PointLight.h:12, AreaLight.h:18-22);
 * the library normalizes normal/u/v as the AreaLight constructor does. */
typedef struct rt_light {
    int32_t type;
    int32_t reserved;
    double intensity[3];
    double position[3]; /* area light: rectangle CENTRE */
    double normal[3];
    double u[3];
    double v[3];
    double width, height;
} rt_light;

/* Scene at the seam (include/Scene.h:17-21). */
typedef struct rt_scene_desc {
    int32_t n_prims;
    int32_t n_materials;
    int32_t n_textures;
    int32_t n_lights;
    const int32_t* prim_type;     /* [n_prims] RT_SPHERE / RT_TRIANGLE / RT_CYLINDER            */
    const int32_t* prim_material; /* [n_prims] index into materials                              */
    const double* prim_geom;      /* [n_prims][RT_GEOM_STRIDE]                                   */
    const rt_material* materials; /* [n_materials]                                               */
    const rt_texture* textures;   /* [n_textures]                                                */
    const rt_light* lights;       /* [n_lights]                                                  */
    double background[3];         /* Scene::backgroundColor                                      */
    int32_t use_bvh;              /* bvhRoot built (RayTracer.cpp:95-99) -> "last leaf wins ties"
                                     (BVHNode.cpp:64-69); 0 -> linear scan, first object wins
                                     (Scene.cpp:72-78)                                           */
    int32_t reserved;
} rt_scene_desc;

/* Camera constants as left by the Camera constructor (src/Camera.cpp:9-31). */
typedef struct rt_camera {
    double position[3];
    double lower_left[3];
    double horizontal[3];
    double vertical[3];
    double u[3];
    double v[3];
    double lens_radius;
} rt_camera;

/* RayTracer scalars (include/RayTracer.h:35-47) plus the partition of one frame over GPUs. */
typedef struct rt_params {
    int32_t width, height;
    int32_t mode;          /* RT_MODE_*                                                        */
    int32_t tonemap;       /* RT_TONEMAP_*                                                     */
    int32_t max_depth;     /* RayTracer::maxDepth ("nbounces")                                 */
    int32_t pixel_samples; /* RayTracer::pixelSamples AS SET, i.e. after the swap at
                              RayTracer.cpp:138-139 (JSON "lightsample")                       */
    int32_t light_samples; /* RayTracer::lightSamples AS SET (JSON "pixelsample")              */
    int32_t reserved0;
    double exposure;
    double shadow_bias; /* 1e-4, include/RayTracer.h:41 */
    uint64_t seed;      /* key of the counter-based RNG (path trace only)                     */
    /* Row-band partition (Whitted/binary and path trace): the frame is cut into bands of
     * band_rows rows; this call renders bands b with b % band_world == band_rank and stores
     * them compactly (local band b / band_world).  band_world = 1 renders the full frame. */
    int32_t band_rows;
    int32_t band_rank;
    int32_t band_world;
    /* Sample-range partition (path trace): strata indices s = sy*sqrt_nspp + sx in
     * [sample_begin, sample_end); sample_end <= 0 means "all". */
    int32_t sample_begin;
    int32_t sample_end;
    int32_t debug_flags; /* 0 in production.  RT_DEBUG_NO_JITTER (path trace): every stratum is sampled at its lower-left
                          * corner instead of a random point (r1 = sx / sqrt_nspp, r2 = sy / sqrt_nspp, RayTracer.cpp:273-274); with
                          * pixelSamples 1 the camera rays are render()'s (u = 1 - i/(W-1), v = j/(H-1), :217-218), which lets the
                          * parity tests drive traceRayPath on the rays of tests/golden/unit_vectors.npz */
} rt_params;
enum { RT_DEBUG_NO_JITTER = 1 };

typedef struct rt_stats {
    uint64_t rays;         /* closest-hit queries traced by the last render (= Scene::intersect
                              calls the reference would make, src/Scene.cpp:63)                */
    uint64_t box_tests;    /* BoundingBox::intersect evaluations actually executed (0 for the
                            * default filter-BVH traversal unless built with -DRT_FBVH_COUNTS=1;
                            * RT_TRAVERSAL=walk / cull always count)                            */
    uint64_t prim_tests;   /* primitive intersect evaluations actually executed               */
    uint64_t sphere_tests; /* ... of which Sphere::intersect                                   */
    uint64_t triangle_tests; /* ... Triangle::intersect                                        */
    uint64_t cylinder_tests; /* ... Cylinder::intersect                                        */
    double kernel_ms;      /* HIP-event time of the last render's kernels on their stream     */
    double upload_ms;      /* last scene upload (host flatten + H2D)                           */
    int32_t n_bvh_nodes;   /* flattened BVH nodes (0 in linear mode)                           */
    int32_t scene_in_lds;  /* 1 if the scene image was staged in LDS                           */
    uint64_t rays_traversed; /* of `rays`, those whose traversal was started: a ray whose answer cannot
                              * reach the image (shadow rays of a hit that refracts, RayTracer.cpp:798;
                              * of a path-traced hit the light cannot reach or see) is counted in `rays`
                              * but not walked.  Counted like box_tests: 0 for the default traversal
                              * unless built with -DRT_FBVH_COUNTS=1                              */
    int32_t kernel_launches; /* launches of the frame's main kernel in the last render (path trace: one per
                              * batch of strata; Whitted: 1): kernel_ms / kernel_launches prices one launch */
    int32_t reserved;
    uint64_t unwalked_ops;   /* RT_TRAVERSAL=walk / cull only: of the box / primitive tests counted above, those of rays
                              * the production traversal does not walk, priced as SURVEY.md 8(d) prices them (24 per box,
                              * 33 / 62 / 150 per sphere / triangle / cylinder test); the roofline's `frac_walked` */
} rt_stats;

typedef struct rt_context rt_context;       /* device-side renderer, one per GPU             */
typedef struct rt_scene_file rt_scene_file; /* host-side result of the JSON ingest           */

/* ---- errors ------------------------------------------------------------------------------ */
const char* rt_last_error(void);
int rt_api_version(void);

/* ---- host ingest: what main() does before the seam (RayTracer.cpp:52-143, 817-1035) ------- */
/* Parses the scene JSON exactly as the reference does (defaults, the cylinder transform, the
 * pixelsample/lightsample swap, texture loading relative to the cwd).  tonemap_name is argv[3]
 * of the reference CLI or NULL. */
int rt_scene_file_load(const char* json_path, const char* tonemap_name, rt_scene_file** out);
void rt_scene_file_free(rt_scene_file* f);
/* Same effect as editing camera.width / camera.height in the JSON and loading again: the image
 * size changes and the camera is rebuilt with the new aspect ratio (RayTracer.cpp:838-840).
 * The shipped scenes are all 1200x800; the 1080p / 4K configurations use this. */
int rt_scene_file_resize(rt_scene_file* f, int32_t width, int32_t height);
const rt_scene_desc* rt_scene_file_desc(const rt_scene_file* f);
const rt_camera* rt_scene_file_camera(const rt_scene_file* f);
const rt_params* rt_scene_file_params(const rt_scene_file* f);

/* Camera::Camera (src/Camera.cpp:9-31). */
int rt_camera_look_at(rt_camera* out, const double position[3], const double look_at[3],
                      const double up[3], double fov_deg, double aspect, double aperture,
                      double focus_dist);

/* RayTracer::writeImageToPPM (RayTracer.cpp:352-367): byte-identical ASCII P3. */
int rt_write_ppm(const char* path, const double* rgb, int32_t width, int32_t height);
/* The same file from the PPM payload bytes of rt_render_u8 / rt_render_device_u8 (already
 * int(255.999 * c), RayTracer.cpp:359-362): byte-identical to rt_write_ppm on the fp64 frame. */
int rt_write_ppm_u8(const char* path, const uint8_t* rgb8, int32_t width, int32_t height);

/* BVHNode::BVHNode (src/BVHNode.cpp:7-45) flattened: returns the number of nodes and, when the
 * out pointers are non-NULL, fills caller arrays of capacity `cap` nodes.  child refs >= 0 are
 * node indices, refs < 0 encode primitive ~ref.  Host only; used by rt_upload_scene and tests.
 *   node_box  [cap][6]  min xyz, max xyz
 *   node_left / node_right [cap]; a leaf (span 1) has left = ~prim and right = INT32_MIN. */
int rt_build_bvh(const rt_scene_desc* scene, int32_t cap, double* node_box, int32_t* node_left,
                 int32_t* node_right);

/* ---- device ------------------------------------------------------------------------------ */
int rt_device_count(void);
int rt_create(rt_context** out, int device);
void rt_destroy(rt_context* ctx);

/* Scene::addObject / addLight / buildBVH (Scene.cpp:15-22, 41-42): copies everything. */
int rt_upload_scene(rt_context* ctx, const rt_scene_desc* scene);
int rt_set_camera(rt_context* ctx, const rt_camera* cam);
int rt_set_params(rt_context* ctx, const rt_params* p);

/* Number of rows / doubles this context's partition produces (after rt_set_params). */
int64_t rt_local_rows(const rt_context* ctx);

/* RayTracer::render / renderPathTrace up to, not including, writeImageToPPM
 * (RayTracer.cpp:206-246, 254-328).  Blocking.  rgb: host, 3*W*local_rows doubles, final
 * colours in [0,1].  hit_id: host, W*local_rows int32 primary-hit primitive IDs, or NULL. */
int rt_render(rt_context* ctx, double* rgb, int32_t* hit_id);

/* Same, asynchronous on `stream` (a hipStream_t, NULL = default stream) into DEVICE buffers
 * the caller owns (e.g. torch tensors that then feed an RCCL collective). */
int rt_render_device(rt_context* ctx, void* stream, double* rgb_dev, int32_t* hit_id_dev);

/* Path trace split across GPUs by sample range: writes the un-normalised radiance SUM of this
 * context's strata (RayTracer.cpp:270-285) to accum_dev (3*W*local_rows doubles), no
 * division, tone map, exposure or gamma: those are non-linear and come after the reduction. */
int rt_pathtrace_accumulate_device(rt_context* ctx, void* stream, double* accum_dev);

/* The component conversion of RayTracer::writeImageToPPM (RayTracer.cpp:359-362), int(255.999 * c),
 * on n_values doubles of a device buffer -> bytes in another device buffer: the PPM's payload, an
 * eighth of the fp64 frame, which is what a multi-GPU job moves over xGMI to the rank that writes
 * the file. */
int rt_quantize_device(rt_context* ctx, void* stream, const double* rgb_dev, int64_t n_values, uint8_t* out_dev);

/* rt_render for callers that only want the image file's payload (the CLI): the frame as
 * int(255.999 * c) bytes in a host buffer of 3*W*local_rows bytes -- an eighth of rt_render's
 * device-to-host copy.  Whitted / binary: written by the kernels' pixel store; path trace:
 * accumulate, resolve (RayTracer.cpp:301-315) and convert on the device. */
int rt_render_u8(rt_context* ctx, uint8_t* rgb8, int32_t* hit_id);

/* rt_render_device with the PPM conversion fused into the kernels' pixel store (binary / phong modes):
 * rgb8_dev receives int(255.999 * c) per component (RayTracer.cpp:359-362 applied to the clamped
 * colour of :231-235), 3 bytes per pixel in the layout of rt_render_device's buffer; rgb_dev, if not
 * null, also receives the fp64 frame.  A job that only writes / ships the PPM thereby skips the fp64
 * frame (24 B per pixel written and read again by rt_quantize_device) and one kernel launch.
 * Path-trace contexts are refused: their frames are sums until rt_pathtrace_resolve_device. */
int rt_render_device_u8(rt_context* ctx, void* stream, uint8_t* rgb8_dev, double* rgb_dev, int32_t* hit_id_dev);

/* rt_render_device_u8 WITHOUT any allocation.  No counterpart in the reference; it exists for callers that cannot
 * call hipMalloc: code under HIP graph capture, tight serving loops, harnesses that forbid allocation.  Same
 * arguments, same frame (binary / phong modes; bit for bit).  All device workspace -- node store, hand-over records,
 * parent lists, counters -- lies in a 256 MiB array that is part of the library's code object (every GPU the library
 * runs on holds its own copy until the process ends); the first call of a (scene, params) setup makes the same
 * attribute / occupancy queries as rt_render_device_u8 but no sizing frame, and nothing grows.  Limits: frames of at
 * most ~3 x 10^5 pixels (about 640 x 480: three node records per pixel must fit; larger frames -> RT_ERR_LIMIT, use
 * rt_render_device_u8); a frame whose ray forest outgrows the array sets the context's overflow flag, reported by
 * the next call on the context or by rt_get_stats as for rt_render_device_u8.  Every call on a GPU shares that GPU's
 * array, whichever context makes it: calls that overlap on one GPU (two streams, two contexts, two threads) race --
 * serialize them.  The pipeline-selection variables of rt_create (RT_PIPELINE, RT_SPLIT_ROOTS) must be unset.
 * Under HIP stream capture: make one call of the setup OUTSIDE capture first (it makes the attribute / occupancy queries);
 * a captured call records no timing events (rt_get_stats then has no kernel_ms / ray count for it) and the overflow word
 * is read by the host at call time, not per replay -- poll it with rt_get_stats after replays. */
int rt_render_device_u8_static(rt_context* ctx, void* stream, uint8_t* rgb8_dev, double* rgb_dev, int32_t* hit_id_dev);

/* Caps the grid of the persistent render kernels at n_workgroups (0 = as many as the chip holds, the
 * default; the environment variable RT_WORKGROUPS sets the initial value).  For callers that keep
 * several frames in flight on one GPU -- one context and one stream per frame, e.g. 4 contexts of 64
 * workgroups: the frames then share the chip's 256 CUs instead of queueing behind one another, which
 * hides the per-level latency of thin frames (a rank's 1/8 share of a 4K frame).  Results do not
 * depend on it.  No counterpart in the reference (its loop is one OpenMP parallel-for,
 * RayTracer.cpp:213-236). */
int rt_set_workgroups(rt_context* ctx, int32_t n_workgroups);

/* RayTracer.cpp:301-315 (path trace) on n_pixels pixels of a device buffer in place:
 * / pixelSamples, tone map, * exposure, pow(1/2.2), clamp. */
int rt_pathtrace_resolve_device(rt_context* ctx, void* stream, double* accum_dev,
                                int64_t n_pixels);

/* Blocks until the last render on this context finished, then fills stats. */
int rt_get_stats(rt_context* ctx, rt_stats* out);

/* ---- several GPUs of one node behind the same seam ------------------------------------------
 * The reference's seam is ONE blocking call (RayTracer.cpp:146-149); a group is what stands behind it when the
 * caller names more than one device ("create context (takes device list)").  One process, one
 * rt_context + stream per device, the scene replicated; a frame is partitioned as SURVEY.md 8(e) lays out --
 * Whitted / binary: 8-row bands dealt round-robin, each member's bands copied into the frame on devices[0] by one
 * strided peer copy over xGMI (as PPM bytes for rt_group_render_u8); path trace: strata ranges, the members' radiance
 * sums combined as a reduce-scatter by peer reads -- member j adds slice j of every member's sums in member order, before
 * the non-linear stages of RayTracer.cpp:301-315, applies those stages to its slice and copies the finished slice to
 * devices[0].  The frame equals the one-device frame (bit for bit for Whitted; to fp summation order, ~1e-16 relative,
 * for path-trace sums).  Every pair of distinct listed devices needs peer access (rt_group_create fails otherwise).
 * The same device may be listed more than once (members then share it).  Pass whole-frame params. */
typedef struct rt_group rt_group;
int rt_group_create(rt_group** out, const int* devices, int n_devices);
void rt_group_destroy(rt_group* g);
int rt_group_size(const rt_group* g);
int rt_group_upload_scene(rt_group* g, const rt_scene_desc* scene);
int rt_group_set_camera(rt_group* g, const rt_camera* cam);
int rt_group_set_params(rt_group* g, const rt_params* p);
/* rt_render / rt_render_u8 for the group: blocking, the whole frame into host buffers (3*W*H doubles / bytes, and
 * optionally W*H primary-hit ids; path-traced frames report -1 as rt_render does not report ids for them either). */
int rt_group_render(rt_group* g, double* rgb, int32_t* hit_id);
int rt_group_render_u8(rt_group* g, uint8_t* rgb8, int32_t* hit_id);
/* Sums of the members' counters; kernel_ms = the slowest member (they render side by side). */
int rt_group_get_stats(rt_group* g, rt_stats* out);

/* ---- diagnostics ---------------------------------------------------------------------------
 * No counterpart in the reference: per-function hooks so that the parity tests can drive
 * Scene::intersect (src/Scene.cpp:63-81) and the device math library on chosen inputs. */
/* Closest hit of n rays given as host arrays [n][3]; directions are normalized as the Ray
 * constructor does (src/Ray.cpp:5-6).  t_out = t or -1, prim_out = primitive ID or -1,
 * normal_out = HitRecord::normal. */
int rt_debug_trace_rays(rt_context* ctx, int64_t n, const double* origins, const double* dirs,
                        double* t_out, int32_t* prim_out, double* normal_out);
/* out[k] = op(a[k], b[k]) evaluated on the device: 0 a/b, 1 sqrt(a), 2 pow(a,b), 3 atan2(a,b),
 * 4 acos(a), 5 cos(a), 6 sin(a), 7 RNG uniform(seed 0; pixel a, sample b, node 1, dim 0),
 * 8 a/b through the shared-reciprocal sequence the traversal uses, 9 / 10 the x (z) component of
 * normalize((a, b, 0.75 b)) (of normalize((b, 0.75 b, a))), 11 / 12 the x (z) component of (a, 0.75 a, -1.5 a) / b
 * (of (0.75 a, -1.5 a, a) / b) through the shared-reciprocal Vector3 / double, 13 sqrt(a) as Vector3::length takes it
 * (the device library's sequence without its rescaling wrapper where that is the identity), 14 a + b + 3a - b summed in the
 * path tracer's exact 128-bit fixed-point accumulator (every term truncated to a multiple of 2^-80, one rounding at the end). */
int rt_debug_math(rt_context* ctx, int op, int64_t n, const double* a, const double* b, double* out);
/* op 0: n rows in[k] = (I xyz, N xyz, eta_t, eta_i) -> out[k] = (refract xyz, 1 if length(normalize(refract)) > 0 else 0
 * [the transmission test of RayTracer.cpp:572], reflect xyz, fresnel, fresnelReflectance(|I.N|, eta_t)), the device's
 * restatement of RayTracer.cpp:430-504.  op 1: n rows of colours (3) -> toneMap for NONE, REINHARD, WARD, UNCHARTED2 (12),
 * RayTracer.cpp:161-202.  op 2: n rows (primitive ID, point xyz) of the uploaded scene -> (u, v) of the primitive's getUV
 * (Sphere.cpp:50-57, Triangle.cpp:15-39, Cylinder.cpp:109-129) and its material's getTextureColor(u, v) (Material.cpp:69-86),
 * 5 values; op 3: n rows (primitive ID, u, v, unused) -> the same with the given (u, v).
 * Checked against the reference's own functions (tests/golden/unit_vectors.npz). */
int rt_debug_optics(rt_context* ctx, int op, int64_t n, const double* in, double* out);

#ifdef __cplusplus
}
#endif
#endif /* RAYTRACER_AMD_H */
