// k_blockwave.hip -- instantiates bw_kernel (blockwave_kernels.inc): the three-phase per-workgroup pipeline, still
// the general path-trace form (any number of lights / samples) and a cross-check for Whitted frames.
#include "rt_device.hpp"

namespace rt {
const void* bw_kernel_ptr(bool pt, bool lds, int tv) {
#define BW_TAB(K)                                                                                        \
    {{{K<0, false, 1>, K<0, false, 2>, K<0, false, 3>}, {K<0, true, 1>, K<0, true, 2>, K<0, true, 3>}},  \
     {{K<1, false, 1>, K<1, false, 2>, K<1, false, 3>}, {K<1, true, 1>, K<1, true, 2>, K<1, true, 3>}}}
    using Fn = void (*)(SceneView, RenderArgs, BwArgs);
    static const Fn tab[2][2][3] = BW_TAB(bw_kernel);
#undef BW_TAB
    return (const void*)tab[pt ? 1 : 0][lds ? 1 : 0][tv];
}
}  // namespace rt
