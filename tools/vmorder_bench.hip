// Micro-benchmark (gfx950): does s_waitcnt vmcnt(N) count loads and stores of one wave IN ORDER?
// A wave issues a load that misses every cache (HBM: ~900+ cycles) and then a store to a line that sits in L2; vmcnt(1) behind the
// two returns when one operation is left outstanding.  In order: that is the store, the wait takes the load's latency.  Out of
// order: the store is done first and the wait returns early, with the load still in flight.
//   hipcc --offload-arch=gfx950 -O3 -o /tmp/vmorder tools/vmorder_bench.hip && /tmp/vmorder
#include <hip/hip_runtime.h>
#include <cstdio>
#include <vector>
typedef float f32x4 __attribute__((ext_vector_type(4)));

__global__ void k(const float4* cold, float4* hot, unsigned long long* out, int mode, size_t stride)
{
    const int lane = threadIdx.x;
    const float4* src = cold + (size_t)blockIdx.x * stride + lane;      // never touched before: HBM
    float4* dst = hot + blockIdx.x * 64 + lane;                          // written in the warm-up: L2
    f32x4 v = {1.f, 2.f, 3.f, 4.f}, ld = {0, 0, 0, 0};
    *(f32x4*)dst = v;                                                            // warm the store's line
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    __builtin_amdgcn_s_sleep(100);
    const unsigned long long t0 = __builtin_amdgcn_s_memtime();
    if (mode == 0) {            // load, store, vmcnt(1)
        asm volatile("global_load_dwordx4 %0, %1, off" : "=v"(ld) : "v"(src) : "memory");
        asm volatile("global_store_dwordx4 %0, %1, off" :: "v"(dst), "v"(v) : "memory");
        asm volatile("s_waitcnt vmcnt(1)" ::: "memory");
    } else if (mode == 1) {     // load, vmcnt(0)
        asm volatile("global_load_dwordx4 %0, %1, off" : "=v"(ld) : "v"(src) : "memory");
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    } else if (mode == 2) {     // store, vmcnt(0)
        asm volatile("global_store_dwordx4 %0, %1, off" :: "v"(dst), "v"(v) : "memory");
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    } else if (mode == 3) {     // store, load, vmcnt(1): the wait is for the store
        asm volatile("global_store_dwordx4 %0, %1, off" :: "v"(dst), "v"(v) : "memory");
        asm volatile("global_load_dwordx4 %0, %1, off" : "=v"(ld) : "v"(src) : "memory");
        asm volatile("s_waitcnt vmcnt(1)" ::: "memory");
    } else if (mode == 4) {     // LDS-DMA load, store, vmcnt(1)
        __shared__ float4 lds[64];
        const unsigned l = (unsigned)(size_t)lds;
        asm volatile("s_mov_b32 m0, %0\n\tglobal_load_lds_dwordx4 %1, off" :: "s"(l), "v"(src) : "memory");
        asm volatile("global_store_dwordx4 %0, %1, off" :: "v"(dst), "v"(v) : "memory");
        asm volatile("s_waitcnt vmcnt(1)" ::: "memory");
        const unsigned long long t1 = __builtin_amdgcn_s_memtime();
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        ld = *(f32x4*)&lds[lane];
        if (lane == 0) out[blockIdx.x] = t1 - t0;
        if (ld.x == 12345.f) out[blockIdx.x] = 0;
        return;
    }
    const unsigned long long t1 = __builtin_amdgcn_s_memtime();
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    if (lane == 0) out[blockIdx.x] = t1 - t0;
    if (ld.x == 12345.f) out[blockIdx.x] = 0;
}

int main()
{
    const int grid = 64;
    const size_t stride = 1 << 20;        // float4s: 16 MB apart
    float4 *cold, *hot; unsigned long long* out;
    hipMalloc(&cold, sizeof(float4) * stride * grid * 6);
    hipMemset(cold, 0, sizeof(float4) * stride * grid * 6);
    hipMalloc(&hot, sizeof(float4) * 64 * grid);
    hipMalloc(&out, 8 * grid);
    const char* name[5] = {"load(HBM), store(L2), vmcnt(1)", "load(HBM), vmcnt(0)", "store(L2), vmcnt(0)", "store(L2), load(HBM), vmcnt(1)", "LDS-DMA(HBM), store(L2), vmcnt(1)"};
    // flush caches between modes by touching another big buffer
    char* flush; hipMalloc(&flush, 1ull << 30);
    for (int mode = 0; mode < 5; ++mode) {
        hipMemset(flush, mode, 1ull << 30);
        hipDeviceSynchronize();
        hipLaunchKernelGGL(k, dim3(grid), dim3(64), 0, 0, cold + (size_t)mode * stride * grid, hot, out, mode, stride);
        hipDeviceSynchronize();
        std::vector<unsigned long long> h(grid);
        hipMemcpy(h.data(), out, 8 * grid, hipMemcpyDeviceToHost);
        unsigned long long mn = ~0ull, mx = 0, sum = 0;
        for (auto x : h) { mn = x < mn ? x : mn; mx = x > mx ? x : mx; sum += x; }
        printf("%-36s ticks until the wait returns: min %llu  mean %llu  max %llu\n", name[mode], mn, sum / grid, mx);
    }
    return 0;
}
