// uva_sww.hip -- translation unit of g_conv3_sww (csrc/uva_sww.hip.h): compiled on its own, like uva_wino.hip.
#include <atomic>

#include "uva_sww.hip.h"

namespace uva {

template <int RES, int RES2, bool RL>
static hipError_t launch_one(hipStream_t stream, int grid, const GSwArgs& a)
{
    auto kfn = g_conv3_sww<RES, RES2, RL>;
    static std::atomic<bool> attr_done[64];       // per device: the kernel's 133 KB of dynamic LDS must be allowed once
    int dev = 0;
    hipError_t e = hipGetDevice(&dev);
    if (e != hipSuccess) return e;
    if (dev < 0 || dev >= 64 || !attr_done[dev].load(std::memory_order_acquire)) {
        e = hipFuncSetAttribute((const void*)kfn, hipFuncAttributeMaxDynamicSharedMemorySize, (int)sww_lds_bytes());
        if (e != hipSuccess) return e;
        if (dev >= 0 && dev < 64) attr_done[dev].store(true, std::memory_order_release);
    }
    hipLaunchKernelGGL(kfn, dim3(grid), dim3(256), sww_lds_bytes(), stream, a);
    return hipGetLastError();
}

hipError_t launch_conv3_sww(hipStream_t stream, int grid, const GSwArgs& a, int res, int res2)
{
    // the first sum's other operand is the convolution's own input (every plane: the same array, the same pixel stride)?
    bool in_ring = res != 0 && a.res_stride == a.in_stride;
    for (int pl = 0; pl < GEN_MAX_PLANES && in_ring; ++pl)
        if (a.in[pl] && a.res[pl] != a.in[pl]) in_ring = false;
    if (res == 0 && res2 == 0) return launch_one<0, 0, false>(stream, grid, a);
    if (res == 2 && res2 == 0) return in_ring ? launch_one<2, 0, true>(stream, grid, a) : launch_one<2, 0, false>(stream, grid, a);
    if (res == 2 && res2 == 2) return in_ring ? launch_one<2, 2, true>(stream, grid, a) : launch_one<2, 2, false>(stream, grid, a);
    return hipErrorInvalidValue;
}

}  // namespace uva
