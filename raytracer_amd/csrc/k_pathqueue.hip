// k_pathqueue.hip -- instantiates pq_kernel (pathqueue_kernels.inc): path tracing out of per-wave queues.
#define RT_TEXTURE_CALL 1  // the textured colour as a real call (rt_device.hpp: textured_colour)
#include "rt_device.hpp"

namespace rt {
const void* pq_kernel_ptr() {
    using Fn = void (*)(SceneView, RenderArgs, PqArgs);
    static const Fn fn = pq_kernel<3>;
    return (const void*)fn;
}
}  // namespace rt
