// k_fused.hip -- instantiates the production Whitted kernels fw_roots_kernel / fw_kernel (fused_kernels.inc).
#include "rt_device.hpp"

namespace rt {
const void* fw_kernel_ptr(bool pt, bool lds, int tv) {
#define BW_TAB(K)                                                                                        \
    {{{K<0, false, 1>, K<0, false, 2>, K<0, false, 3>}, {K<0, true, 1>, K<0, true, 2>, K<0, true, 3>}},  \
     {{K<1, false, 1>, K<1, false, 2>, K<1, false, 3>}, {K<1, true, 1>, K<1, true, 2>, K<1, true, 3>}}}
    using Fn = void (*)(SceneView, RenderArgs, BwArgs);
    static const Fn tab[2][2][3] = BW_TAB(fw_kernel);
#undef BW_TAB
    return (const void*)tab[pt ? 1 : 0][lds ? 1 : 0][tv];
}
const void* fw_roots_kernel_ptr(bool lds, int tv) {
    using Fn = void (*)(SceneView, RenderArgs, BwArgs);
    static const Fn tab[2][3] = {{fw_roots_kernel<false, 1>, fw_roots_kernel<false, 2>, fw_roots_kernel<false, 3>},
                                 {fw_roots_kernel<true, 1>, fw_roots_kernel<true, 2>, fw_roots_kernel<true, 3>}};
    return (const void*)tab[lds ? 1 : 0][tv];
}
}  // namespace rt
