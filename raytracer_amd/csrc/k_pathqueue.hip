// k_pathqueue.hip -- instantiates pq_kernel (pathqueue_kernels.inc): path tracing out of per-wave queues.
#define RT_TEXTURE_CALL 1  // the textured colour as a real call (rt_device.hpp: textured_colour)
#include "rt_device.hpp"

namespace rt {
const void* pq_kernel_ptr(int shadow_rays_per_hit) {
    using Fn = void (*)(SceneView, RenderArgs, PqArgs);
    static const Fn s16 = pq_kernel<3, 16>, any = pq_kernel<3, 0>;
    return (const void*)(shadow_rays_per_hit == 16 ? s16 : any);
}
}  // namespace rt
