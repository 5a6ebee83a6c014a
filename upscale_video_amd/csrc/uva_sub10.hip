// uva_sub10.hip -- translation unit of sub10_kernel (csrc/uva_sub10.hip.h): compiled on its own, like uva_wino.hip.
#include <atomic>

#include "uva_sub10.hip.h"

namespace uva {

hipError_t launch_sub10_kernel(hipStream_t stream, int grid, const Sub10Args& a)
{
    static std::atomic<bool> attr_done[64];       // per device: the kernel's 160 KB of dynamic LDS must be allowed once
    int dev = 0;
    hipError_t e = hipGetDevice(&dev);
    if (e != hipSuccess) return e;
    if (dev < 0 || dev >= 64 || !attr_done[dev].load(std::memory_order_acquire)) {
        e = hipFuncSetAttribute((const void*)sub10_kernel, hipFuncAttributeMaxDynamicSharedMemorySize, (int)sub10_lds_bytes());
        if (e != hipSuccess) return e;
        if (dev >= 0 && dev < 64) attr_done[dev].store(true, std::memory_order_release);
    }
    hipLaunchKernelGGL(sub10_kernel, dim3(grid), dim3(64 * S10_NW), sub10_lds_bytes(), stream, a);
    return hipGetLastError();
}

}  // namespace uva
