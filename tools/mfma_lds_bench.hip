// Microbenchmark 2: MFMA issue rate with the k-loop's operand traffic: 36 distinct A fragments in
// registers, B fragments streamed from LDS by ds_read_b128 PF steps ahead (1 wave per SIMD).
#include <hip/hip_runtime.h>
#include <cstdio>
typedef _Float16 half8 __attribute__((ext_vector_type(8)));
typedef float f32x16 __attribute__((ext_vector_type(16)));

template <int MODE>   // 0: B from LDS ring; 1: B constant in registers; 2: LDS reads issued but B constant
__global__ __launch_bounds__(256, 1) void k(const half8* in, float* out, long long* cyc)
{
    __shared__ __attribute__((aligned(16))) char lds[32768];
    for (int i = threadIdx.x; i < 2048; i += 256) ((half8*)lds)[i] = in[i & 511];
    half8 w[36];
#pragma unroll
    for (int i = 0; i < 36; ++i) w[i] = in[(i * 7 + threadIdx.x) & 511];
    __syncthreads();
    const char* base = lds + (threadIdx.x & 63) * 144;
    f32x16 acc0, acc1;
#pragma unroll
    for (int r = 0; r < 16; ++r) { acc0[r] = 0.f; acc1[r] = 0.f; }
    half8 cb = in[threadIdx.x];
    long long t0 = __builtin_amdgcn_s_memtime();
    for (int rep = 0; rep < 64; ++rep) {
        constexpr int PF = 6;
        half8 bq[PF + 1];
#pragma unroll
        for (int s = 0; s < PF; ++s) bq[s] = *(const half8*)(base + s * 32);
#pragma unroll
        for (int s = 0; s < 72; ++s) {
            if (MODE != 1 && s + PF < 72) bq[(s + PF) % (PF + 1)] = *(const half8*)(base + ((s + PF) * 32) % 20000);
            half8 b = (MODE == 0) ? bq[s % (PF + 1)] : cb;
            if (MODE == 2) asm volatile("" ::"v"(bq[s % (PF + 1)]));
            if (s & 1) acc1 = __builtin_amdgcn_mfma_f32_32x32x16_f16(w[s % 36], b, acc1, 0, 0, 0);
            else acc0 = __builtin_amdgcn_mfma_f32_32x32x16_f16(w[s % 36], b, acc0, 0, 0, 0);
        }
        __builtin_amdgcn_sched_group_barrier(0x100, PF, 0);
#pragma unroll
        for (int s = 0; s < 72; ++s) {
            __builtin_amdgcn_sched_group_barrier(0x008, 1, 0);
            if (MODE != 1 && s + PF < 72) __builtin_amdgcn_sched_group_barrier(0x100, 1, 0);
        }
    }
    long long t1 = __builtin_amdgcn_s_memtime();
    out[blockIdx.x * 256 + threadIdx.x] = acc0[0] + acc1[5];
    if (threadIdx.x == 0 && blockIdx.x == 0) *cyc = t1 - t0;
}

template <typename K>
void run(const char* name, K kf, half8* in, float* out, long long* cyc)
{
    for (int r = 0; r < 2; ++r) { hipLaunchKernelGGL(kf, dim3(256), dim3(256), 0, 0, in, out, cyc); (void)hipDeviceSynchronize(); }
    long long c = 0;
    (void)hipMemcpy(&c, cyc, 8, hipMemcpyDeviceToHost);
    printf("%-44s %7.2f ticks per MFMA\n", name, (double)c / (64.0 * 72));
}

int main()
{
    half8* in; float* out; long long* cyc;
    (void)hipMalloc(&in, 512 * 16); (void)hipMemset(in, 0x3c, 512 * 16);
    (void)hipMalloc(&out, 256 * 256 * 4); (void)hipMalloc(&cyc, 8);
    run("B from LDS ring (ds_read_b128, PF=6)", k<0>, in, out, cyc);
    run("B constant, no LDS reads", k<1>, in, out, cyc);
    run("B constant, LDS reads issued and waited", k<2>, in, out, cyc);
    return 0;
}
