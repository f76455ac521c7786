// k_pathwave.hip -- instantiates the production path-trace kernel pt_kernel (pathwave_kernels.inc).
#define RT_TEXTURE_CALL 1  // the textured colour as a real call (rt_device.hpp: textured_colour)
#include "rt_device.hpp"

namespace rt {
const void* pt_kernel_ptr(bool lds, int tv) {
    using Fn = void (*)(SceneView, RenderArgs, BwArgs);
    static const Fn tab[2][3] = {{pt_kernel<false, 1>, pt_kernel<false, 2>, pt_kernel<false, 3>},
                                 {pt_kernel<true, 1>, pt_kernel<true, 2>, pt_kernel<true, 3>}};
    return (const void*)tab[lds ? 1 : 0][tv];
}
}  // namespace rt
