// k_pathqueue.hip -- instantiates pq_kernel (pathqueue_kernels.inc): path tracing out of per-wave queues.
#define RT_TEXTURE_CALL 1  // the textured colour as a real call (rt_device.hpp: textured_colour)
#include "rt_device.hpp"

namespace rt {
const void* pq_kernel_ptr(int shadow_rays_per_hit, int waves) {
    using Fn = void (*)(SceneView, RenderArgs, PqArgs);
    static const Fn tab[2][2] = {{pq_kernel<3, 0, kPqWavesFew>, pq_kernel<3, 0, kPqWavesMany>},
                                 {pq_kernel<3, 16, kPqWavesFew>, pq_kernel<3, 16, kPqWavesMany>}};
    return (const void*)tab[shadow_rays_per_hit == 16 ? 1 : 0][waves == kPqWavesMany ? 1 : 0];
}
}  // namespace rt
