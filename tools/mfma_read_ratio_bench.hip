// One wave per SIMD (a 512-register kernel's situation): how much of a ds_read_b128's time hides behind MFMAs, for the
// two fp16 MFMA shapes at the same flop per cycle and the same LDS bytes per flop?
//   SHAPE 16: v_mfma_f32_16x16x32_f16 (16 cycles each);  SHAPE 32: v_mfma_f32_32x32x16_f16 (32 cycles each)
//   R = fragment reads (1 KiB per wave) per 64 MFMA-cycles: 0, 1 (= 0.25 reads per 16x16x32 MFMA), 2 (= 0.5), 4 (= 1)
// Every read result feeds a later MFMA's B operand (a true dependency, PF groups ahead); reads are conflict-free
// (lane * 16 inside a rotating 1-KiB window).  Prints shader cycles per 64 MFMA-cycles of work (s_memtime, workgroup 0)
// and the whole chip's TFLOP/s.
// build: hipcc --offload-arch=gfx950 -O3 tools/mfma_read_ratio_bench.hip -o /tmp/mrr && /tmp/mrr
#include <hip/hip_runtime.h>
#include <cstdio>
#include <vector>
typedef _Float16 half8 __attribute__((ext_vector_type(8)));
typedef float f32x4 __attribute__((ext_vector_type(4)));
typedef float f32x16 __attribute__((ext_vector_type(16)));

constexpr int GROUPS = 64;        // groups of 64 MFMA-cycles per loop body
constexpr int REPS = 400;

template <int SHAPE, int R, int NW>
__global__ __launch_bounds__(256, 1) void k(const half8* in, float* out, long long* cyc)
{
    extern __shared__ __attribute__((aligned(16))) char lds[];
    for (int i = threadIdx.x; i < 9000; i += 256) ((half8*)lds)[i] = in[i & 511];
    half8 w[NW];                                // NW distinct A fragments: beyond ~50 they live partly in AccVGPRs
#pragma unroll
    for (int i = 0; i < NW; ++i) w[i] = in[(i * 7 + threadIdx.x) & 511];
    __syncthreads();
    const char* base = lds + (threadIdx.x & 63) * 16 + (threadIdx.x >> 6) * 1024;
    constexpr int PF = 3;                       // groups the reads run ahead
    constexpr int NR = R ? R : 1;
    half8 bq[PF + 1][NR];
#pragma unroll
    for (int s = 0; s <= PF; ++s)
#pragma unroll
        for (int r = 0; r < NR; ++r) bq[s][r] = in[(s * 5 + r + threadIdx.x) & 511];
    f32x16 a32[2];
    f32x4 a16[4];
#pragma unroll
    for (int r = 0; r < 16; ++r) { a32[0][r] = 0.f; a32[1][r] = 0.f; }
#pragma unroll
    for (int c = 0; c < 4; ++c)
#pragma unroll
        for (int r = 0; r < 4; ++r) a16[c][r] = 0.f;
    const long long t0 = __builtin_amdgcn_s_memtime();
    for (int rep = 0; rep < REPS; ++rep) {
        __builtin_amdgcn_sched_barrier(0);
#pragma unroll
        for (int g = 0; g < GROUPS; ++g) {
            if constexpr (R > 0) {
#pragma unroll
                for (int r = 0; r < R; ++r) bq[(g + PF) % (PF + 1)][r] = *(const half8*)(base + (g * R + r) * 512);     // (all distinct: nothing to merge)
            }
            if constexpr (SHAPE == 16) {
#pragma unroll
                for (int i = 0; i < 4; ++i) a16[i] = __builtin_amdgcn_mfma_f32_16x16x32_f16(w[(g * (SHAPE == 16 ? 4 : 2) + i) % NW], bq[g % (PF + 1)][i % NR], a16[i], 0, 0, 0);
            } else {
#pragma unroll
                for (int i = 0; i < 2; ++i) a32[i] = __builtin_amdgcn_mfma_f32_32x32x16_f16(w[(g * (SHAPE == 16 ? 4 : 2) + i) % NW], bq[g % (PF + 1)][i % NR], a32[i], 0, 0, 0);
            }
            if constexpr (R > 0) __builtin_amdgcn_sched_group_barrier(0x100, R, 0);
            __builtin_amdgcn_sched_group_barrier(0x008, SHAPE == 16 ? 4 : 2, 0);
        }
        __builtin_amdgcn_sched_barrier(0);
    }
    const long long t1 = __builtin_amdgcn_s_memtime();
    float s = 0;
#pragma unroll
    for (int c = 0; c < 4; ++c) s += a16[c][0] + a16[c][3];
    s += a32[0][0] + a32[1][15];
    out[blockIdx.x * 256 + threadIdx.x] = s;
    if (blockIdx.x == 0 && threadIdx.x == 0) cyc[0] = t1 - t0;
}

template <int SHAPE, int R, int NW = 8>
void run(const half8* din, float* dout, long long* dcyc, int ncu)
{
    hipFuncSetAttribute((const void*)k<SHAPE, R, NW>, hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024);
    hipEvent_t e0, e1;
    hipEventCreate(&e0); hipEventCreate(&e1);
    hipLaunchKernelGGL((k<SHAPE, R, NW>), dim3(ncu), dim3(256), 160 * 1024 - 256, 0, din, dout, dcyc);
    hipDeviceSynchronize();
    hipEventRecord(e0);
    for (int i = 0; i < 10; ++i) hipLaunchKernelGGL((k<SHAPE, R, NW>), dim3(ncu), dim3(256), 160 * 1024 - 256, 0, din, dout, dcyc);
    hipEventRecord(e1);
    hipEventSynchronize(e1);
    float ms; hipEventElapsedTime(&ms, e0, e1);
    long long c; hipMemcpy(&c, dcyc, 8, hipMemcpyDeviceToHost);
    const double flop = 10.0 * ncu * 4 * (double)REPS * GROUPS * 64 * 1024;       // 1024 flop per cycle and SIMD
    printf("shape %2d  A fragments %2d  reads per 64 MFMA-cycles %d : %6.1f cycles per 64 MFMA-cycles (workgroup 0), %7.1f TFLOP/s\n", SHAPE, NW, R,
           (double)c / ((double)REPS * GROUPS), flop / (ms * 1e-3) / 1e12);
}

int main()
{
    hipDeviceProp_t p; hipGetDeviceProperties(&p, 0);
    const int ncu = p.multiProcessorCount;
    std::vector<_Float16> h(512 * 8);
    for (size_t i = 0; i < h.size(); ++i) h[i] = (_Float16)(((int)(i * 2654435761u >> 20) % 200 - 100) / 400.0f);
    half8* din; float* dout; long long* dcyc;
    hipMalloc(&din, h.size() * 2); hipMalloc(&dout, (size_t)ncu * 256 * 4); hipMalloc(&dcyc, 8);
    hipMemcpy(din, h.data(), h.size() * 2, hipMemcpyHostToDevice);
    run<16, 0>(din, dout, dcyc, ncu); run<16, 1>(din, dout, dcyc, ncu); run<16, 2>(din, dout, dcyc, ncu); run<16, 4>(din, dout, dcyc, ncu);
    run<32, 0>(din, dout, dcyc, ncu); run<32, 1>(din, dout, dcyc, ncu); run<32, 2>(din, dout, dcyc, ncu); run<32, 4>(din, dout, dcyc, ncu);
    run<16, 0, 54>(din, dout, dcyc, ncu); run<16, 1, 54>(din, dout, dcyc, ncu); run<16, 2, 54>(din, dout, dcyc, ncu);
    run<16, 2, 72>(din, dout, dcyc, ncu); run<32, 2, 54>(din, dout, dcyc, ncu); run<32, 4, 54>(din, dout, dcyc, ncu);
    run<16, 0, 100>(din, dout, dcyc, ncu); run<16, 1, 100>(din, dout, dcyc, ncu); run<16, 2, 100>(din, dout, dcyc, ncu); run<32, 2, 100>(din, dout, dcyc, ncu);
    return 0;
}
