// uva_wino.hip -- translation unit of trunkw_kernel (csrc/uva_wino.hip.h): compiled on its own, so that the kernel can
// be rebuilt in seconds while the rest of libuva.so takes minutes.
#include <atomic>

#include "uva_wino.hip.h"

namespace uva {

template <int ACT>
static hipError_t launch_act(hipStream_t stream, int grid, const TrunkwArgs& a)
{
    auto kfn = trunkw_kernel<64, ACT>;
    static std::atomic<bool> attr_done[64];        // per device: the kernel's 158.5 KB of dynamic LDS must be allowed once
    int dev = 0;
    hipError_t e = hipGetDevice(&dev);
    if (e != hipSuccess) return e;
    if (dev < 0 || dev >= 64 || !attr_done[dev].load(std::memory_order_acquire)) {
        e = hipFuncSetAttribute((const void*)kfn, hipFuncAttributeMaxDynamicSharedMemorySize, (int)TW_LDS_BYTES);
        if (e != hipSuccess) return e;
        if (dev >= 0 && dev < 64) attr_done[dev].store(true, std::memory_order_release);
    }
    hipLaunchKernelGGL(kfn, dim3(grid), dim3(512), TW_LDS_BYTES, stream, a);
    return hipGetLastError();
}

hipError_t launch_trunkw_kernel(hipStream_t stream, int grid, const TrunkwArgs& a, int act)
{
    switch (act) {
    case TW_ACT_F32: return launch_act<TW_ACT_F32>(stream, grid, a);
    case TW_ACT_F16: return launch_act<TW_ACT_F16>(stream, grid, a);
    case TW_ACT_F16_FLIP: return launch_act<TW_ACT_F16_FLIP>(stream, grid, a);
    }
    return hipErrorInvalidValue;
}

}  // namespace uva
