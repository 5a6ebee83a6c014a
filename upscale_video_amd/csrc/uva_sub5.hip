// uva_sub5.hip -- translation unit of sub5_kernel (csrc/uva_sub5.hip.h): compiled on its own, like uva_wino.hip.
#include <atomic>

#include "uva_sub5.hip.h"

namespace uva {

template <int PART>
static hipError_t launch_part(hipStream_t stream, int grid, const Sub5Args& a)
{
    auto kfn = sub5_kernel<PART>;
    static std::atomic<bool> attr_done[64];        // per device: the kernel's 138 KB of dynamic LDS must be allowed once
    int dev = 0;
    hipError_t e = hipGetDevice(&dev);
    if (e != hipSuccess) return e;
    if (dev < 0 || dev >= 64 || !attr_done[dev].load(std::memory_order_acquire)) {
        e = hipFuncSetAttribute((const void*)kfn, hipFuncAttributeMaxDynamicSharedMemorySize, (int)s5::LDS_BYTES);
        if (e != hipSuccess) return e;
        if (dev >= 0 && dev < 64) attr_done[dev].store(true, std::memory_order_release);
    }
    hipLaunchKernelGGL(kfn, dim3(grid), dim3(64 * s5::NW), s5::LDS_BYTES, stream, a);
    return hipGetLastError();
}

hipError_t launch_sub5_kernel(hipStream_t stream, int grid, const Sub5Args& a, int part)
{
    return part == 0 ? launch_part<0>(stream, grid, a) : part == 1 ? launch_part<1>(stream, grid, a) : hipErrorInvalidValue;
}

}  // namespace uva
