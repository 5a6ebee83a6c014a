// Sustained matrix-pipe throughput of the whole chip under its power cap: every SIMD of every CU
// issues back-to-back v_mfma_f32_32x32x16_f16 on register operands (no LDS, no memory traffic) for
// several seconds.  The result is the roof a kernel can reach at the package power limit -- on
// MI355X well below 256 CUs x 4 SIMDs x 1024 flop/cycle x 2.4 GHz, because the clock drops to keep
// the package at 1400 W.  Operands: 0 = zeros, 1 = random fp16 (data toggling costs power).
// build: hipcc --offload-arch=gfx950 -O3 tools/mfma_power_bench.hip -o /tmp/mfma_power_bench
// run:   /tmp/mfma_power_bench [seconds=8]
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#include <vector>
typedef _Float16 half8 __attribute__((ext_vector_type(8)));
typedef float f32x16 __attribute__((ext_vector_type(16)));

constexpr int REPS = 4096, UNROLL = 16;

// SHAPE 32: v_mfma_f32_32x32x16_f16 (2 chains);  SHAPE 16: v_mfma_f32_16x16x32_f16 (4 chains, same
// flop per cycle).  PAT 0: A and B operands change from one MFMA to the next; 1: A fixed, B changes;
// 2: B fixed, A changes; 3: both fixed.
typedef float f32x4 __attribute__((ext_vector_type(4)));
template <int WAVES, int SHAPE, int PAT>
__global__ __launch_bounds__(WAVES * 64, 1) void burn(const half8* in, float* out)
{
    half8 a[4], b[4];
#pragma unroll
    for (int i = 0; i < 4; ++i) {
        a[i] = in[(2 * i) * 512 + threadIdx.x % 512];
        b[i] = in[(2 * i + 1) * 512 + threadIdx.x % 512];
    }
    float s = 0;
    if constexpr (SHAPE == 32) {
        f32x16 acc[2];
#pragma unroll
        for (int c = 0; c < 2; ++c)
#pragma unroll
            for (int r = 0; r < 16; ++r) acc[c][r] = 0.f;
        for (int rep = 0; rep < REPS; ++rep) {
#pragma unroll
            for (int i = 0; i < UNROLL; ++i)
                acc[i & 1] = __builtin_amdgcn_mfma_f32_32x32x16_f16(a[(PAT & 1) ? 0 : (i & 3)], b[(PAT & 2) ? 0 : ((i >> 1) & 3)],
                                                                    acc[i & 1], 0, 0, 0);
        }
#pragma unroll
        for (int c = 0; c < 2; ++c) s += acc[c][0] + acc[c][15];
    } else {
        f32x4 acc[4];
#pragma unroll
        for (int c = 0; c < 4; ++c)
#pragma unroll
            for (int r = 0; r < 4; ++r) acc[c][r] = 0.f;
        for (int rep = 0; rep < 2 * REPS; ++rep) {
#pragma unroll
            for (int i = 0; i < UNROLL; ++i)
                acc[i & 3] = __builtin_amdgcn_mfma_f32_16x16x32_f16(a[(PAT & 1) ? 0 : (i & 3)], b[(PAT & 2) ? 0 : ((i >> 2) & 3)],
                                                                    acc[i & 3], 0, 0, 0);
        }
#pragma unroll
        for (int c = 0; c < 4; ++c) s += acc[c][0] + acc[c][3];
    }
    out[blockIdx.x * WAVES * 64 + threadIdx.x] = s;
}

template <int WAVES, int SHAPE, int PAT>
static void run(const char* name, const half8* in, float* out, double seconds)
{
    hipEvent_t e0, e1;
    hipEventCreate(&e0);
    hipEventCreate(&e1);
    const double flop = 256.0 * WAVES * REPS * UNROLL * 32768.0;
    double elapsed = 0;
    while (elapsed < seconds) {
        hipEventRecord(e0);
        for (int i = 0; i < 20; ++i) hipLaunchKernelGGL((burn<WAVES, SHAPE, PAT>), dim3(256), dim3(WAVES * 64), 0, 0, in, out);
        hipEventRecord(e1);
        hipEventSynchronize(e1);
        float ms;
        hipEventElapsedTime(&ms, e0, e1);
        elapsed += ms * 1e-3;
        if (elapsed >= seconds)
            printf("%-44s %.1f TFLOP/s  (=> %.0f MHz if the pipe never idles)\n", name,
                   20 * flop / (ms * 1e-3) * 1e-12, 20 * flop / (ms * 1e-3) / (1024.0 * 4 * 256) * 1e-6);
    }
}

int main(int argc, char** argv)
{
    const double seconds = argc > 1 ? atof(argv[1]) : 8.0;
    half8* in;
    float* out;
    hipMalloc(&in, 8 * 512 * sizeof(half8));
    hipMalloc(&out, 256 * 512 * sizeof(float));
    for (int mode = 0; mode < 3; ++mode) {
        // 0 zeros; 1 uniform random in [-1,1); 2 small-magnitude values like trained weights / PReLU outputs
        std::vector<_Float16> h(8 * 512 * 8);
        srand(1);
        for (auto& v : h) {
            const float r = rand() / (float)RAND_MAX - 0.5f;
            v = mode == 0 ? (_Float16)0.f : mode == 1 ? (_Float16)(2.f * r) : (_Float16)(0.05f * r * r * r * 8.f);
        }
        hipMemcpy(in, h.data(), h.size() * 2, hipMemcpyHostToDevice);
        const char* d = mode == 0 ? "zeros " : mode == 1 ? "random" : "small ";
        char nm[128];
        snprintf(nm, sizeof nm, "%s 32x32x16 1 wave/SIMD, A,B change", d);  run<4, 32, 0>(nm, in, out, seconds);
        if (mode == 0) { snprintf(nm, sizeof nm, "%s 32x32x16 2 waves/SIMD", d); run<8, 32, 0>(nm, in, out, seconds); }
        if (mode == 0) continue;
        snprintf(nm, sizeof nm, "%s 32x32x16 1 wave/SIMD, A fixed", d);      run<4, 32, 1>(nm, in, out, seconds);
        snprintf(nm, sizeof nm, "%s 32x32x16 1 wave/SIMD, B fixed", d);      run<4, 32, 2>(nm, in, out, seconds);
        snprintf(nm, sizeof nm, "%s 32x32x16 1 wave/SIMD, A,B fixed", d);    run<4, 32, 3>(nm, in, out, seconds);
        snprintf(nm, sizeof nm, "%s 16x16x32 1 wave/SIMD, A,B change", d);   run<4, 16, 0>(nm, in, out, seconds);
        snprintf(nm, sizeof nm, "%s 16x16x32 1 wave/SIMD, A fixed", d);      run<4, 16, 1>(nm, in, out, seconds);
        snprintf(nm, sizeof nm, "%s 32x32x16 2 waves/SIMD, A,B change", d);  run<8, 32, 0>(nm, in, out, seconds);
    }
    return 0;
}
