// Microbenchmark: issue rate of v_mfma_f32_32x32x16_f16 / 16x16x32 as a function of how many
// independent accumulator chains one wave interleaves (1 wave per SIMD, 256 threads per CU).
// build: hipcc --offload-arch=gfx950 -O3 tools/mfma_chain_bench.hip -o /tmp/mfma_chain_bench
#include <hip/hip_runtime.h>
#include <cstdio>
typedef _Float16 half8 __attribute__((ext_vector_type(8)));
typedef float f32x16 __attribute__((ext_vector_type(16)));
typedef float f32x4 __attribute__((ext_vector_type(4)));

template <int CHAINS, int N>
__global__ __launch_bounds__(256, 1) void bench32(const half8* in, float* out, long long* cyc)
{
    half8 a = in[threadIdx.x], b = in[256 + threadIdx.x];
    f32x16 acc[CHAINS];
#pragma unroll
    for (int c = 0; c < CHAINS; ++c)
#pragma unroll
        for (int r = 0; r < 16; ++r) acc[c][r] = 0.f;
    __syncthreads();
    long long t0 = __builtin_amdgcn_s_memtime();
    for (int rep = 0; rep < 64; ++rep) {
#pragma unroll
        for (int i = 0; i < N; ++i) acc[i % CHAINS] = __builtin_amdgcn_mfma_f32_32x32x16_f16(a, b, acc[i % CHAINS], 0, 0, 0);
    }
    long long t1 = __builtin_amdgcn_s_memtime();
    float s = 0;
#pragma unroll
    for (int c = 0; c < CHAINS; ++c) s += acc[c][0] + acc[c][15];
    out[blockIdx.x * 256 + threadIdx.x] = s;
    if (threadIdx.x == 0 && blockIdx.x == 0) *cyc = t1 - t0;
}

template <int CHAINS, int N>
__global__ __launch_bounds__(256, 1) void bench16(const half8* in, float* out, long long* cyc)
{
    half8 a = in[threadIdx.x], b = in[256 + threadIdx.x];
    f32x4 acc[CHAINS];
#pragma unroll
    for (int c = 0; c < CHAINS; ++c)
#pragma unroll
        for (int r = 0; r < 4; ++r) acc[c][r] = 0.f;
    __syncthreads();
    long long t0 = __builtin_amdgcn_s_memtime();
    for (int rep = 0; rep < 64; ++rep) {
#pragma unroll
        for (int i = 0; i < N; ++i) acc[i % CHAINS] = __builtin_amdgcn_mfma_f32_16x16x32_f16(a, b, acc[i % CHAINS], 0, 0, 0);
    }
    long long t1 = __builtin_amdgcn_s_memtime();
    float s = 0;
#pragma unroll
    for (int c = 0; c < CHAINS; ++c) s += acc[c][0] + acc[c][3];
    out[blockIdx.x * 256 + threadIdx.x] = s;
    if (threadIdx.x == 0 && blockIdx.x == 0) *cyc = t1 - t0;
}

template <typename K>
void run(const char* name, K k, int n, half8* in, float* out, long long* cyc)
{
    hipLaunchKernelGGL(k, dim3(256), dim3(256), 0, 0, in, out, cyc);
    hipDeviceSynchronize();
    hipEvent_t e0, e1;
    hipEventCreate(&e0); hipEventCreate(&e1);
    hipEventRecord(e0);
    hipLaunchKernelGGL(k, dim3(256), dim3(256), 0, 0, in, out, cyc);
    hipEventRecord(e1);
    hipDeviceSynchronize();
    long long c = 0; float ms = 0;
    hipMemcpy(&c, cyc, 8, hipMemcpyDeviceToHost);
    hipEventElapsedTime(&ms, e0, e1);
    printf("%-28s %7.2f s_memtime ticks per MFMA   (kernel %.1f us)\n", name, (double)c / (64.0 * n), ms * 1e3);
}

int main()
{
    half8* in; float* out; long long* cyc;
    hipMalloc(&in, 512 * 16); hipMemset(in, 0x3c, 512 * 16);
    hipMalloc(&out, 256 * 256 * 4); hipMalloc(&cyc, 8);
    run("32x32x16 1 chain", bench32<1, 72>, 72, in, out, cyc);
    run("32x32x16 2 chains", bench32<2, 72>, 72, in, out, cyc);
    run("32x32x16 3 chains", bench32<3, 72>, 72, in, out, cyc);
    run("32x32x16 4 chains", bench32<4, 72>, 72, in, out, cyc);
    run("16x16x32 1 chain", bench16<1, 144>, 144, in, out, cyc);
    run("16x16x32 2 chains", bench16<2, 144>, 144, in, out, cyc);
    run("16x16x32 4 chains", bench16<4, 144>, 144, in, out, cyc);
    run("16x16x32 8 chains", bench16<8, 144>, 144, in, out, cyc);
    return 0;
}
