// Microbenchmark 3: issue cost of global_load_lds_dwordx4 (LDS-DMA) pieces, 4 waves per CU.
#include <hip/hip_runtime.h>
#include <cstdio>
template <int MODE>   // 0: m0 save/restore per piece (as in the kernels); 1: m0 set once per piece, no restore
__global__ __launch_bounds__(256, 1) void k(const char* src, float* out, long long* cyc, int stride)
{
    extern __shared__ __attribute__((aligned(16))) char lds[];
    const int wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6), lane = threadIdx.x & 63;
    const char* base = src + (size_t)blockIdx.x * 65536;
    long long t0 = __builtin_amdgcn_s_memtime();
    for (int rep = 0; rep < 8; ++rep) {
#pragma unroll
        for (int i = 0; i < 8; ++i) {
            const unsigned voff = (unsigned)((rep * 8 + i) * 1024 + (lane / 9) * stride + (lane % 9) * 16);
            const unsigned dst = (wave * 8 + i) * 1024;
            if (MODE == 0) {
                unsigned keep;
                asm volatile("s_mov_b32 %0, m0\n\ts_mov_b32 m0, %3\n\ts_nop 0\n\tglobal_load_lds_dwordx4 %1, %2\n\ts_mov_b32 m0, %0"
                             : "=&s"(keep) : "v"(voff), "s"(base), "s"(dst));
            } else {
                asm volatile("s_mov_b32 m0, %2\n\ts_nop 0\n\tglobal_load_lds_dwordx4 %0, %1" ::"v"(voff), "s"(base), "s"(dst) : "m0");
            }
        }
    }
    long long t1 = __builtin_amdgcn_s_memtime();
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    long long t2 = __builtin_amdgcn_s_memtime();
    out[blockIdx.x * 256 + threadIdx.x] = lds[threadIdx.x];
    if (threadIdx.x == 0 && blockIdx.x == 0) { cyc[0] = t1 - t0; cyc[1] = t2 - t0; }
}
template <typename K>
void run(const char* name, K kf, char* src, float* out, long long* cyc, int stride)
{
    for (int r = 0; r < 2; ++r) { hipLaunchKernelGGL(kf, dim3(256), dim3(256), 65536, 0, src, out, cyc, stride); (void)hipDeviceSynchronize(); }
    long long c[2];
    (void)hipMemcpy(c, cyc, 16, hipMemcpyDeviceToHost);
    printf("%-40s issue %6.1f ticks/piece, issue+landed %6.1f ticks/piece\n", name, c[0] / 64.0, c[1] / 64.0);
}
int main()
{
    char* src; float* out; long long* cyc;
    (void)hipMalloc(&src, 256ull * 65536 + 1048576); (void)hipMemset(src, 1, 256ull * 65536 + 1048576);
    (void)hipMalloc(&out, 256 * 256 * 4); (void)hipMalloc(&cyc, 16);
    (void)hipFuncSetAttribute((const void*)k<0>, hipFuncAttributeMaxDynamicSharedMemorySize, 65536);
    (void)hipFuncSetAttribute((const void*)k<1>, hipFuncAttributeMaxDynamicSharedMemorySize, 65536);
    run("m0 save/restore, pixel stride 128", k<0>, src, out, cyc, 128);
    run("m0 clobber, pixel stride 128", k<1>, src, out, cyc, 128);
    run("m0 clobber, contiguous (stride 144)", k<1>, src, out, cyc, 144);
    return 0;
}
