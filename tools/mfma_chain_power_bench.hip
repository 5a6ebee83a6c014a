// How the ORDER of accumulators affects what v_mfma_f32_16x16x32_f16 sustains at the package power cap:
// CH independent accumulator chains visited round-robin with RUN consecutive MFMAs on the same accumulator
// (RUN > 1: back-to-back dependent MFMAs), A fixed for AF consecutive MFMAs.  Random fp16 operands, all 1024
// SIMDs, one wave per SIMD, sustained for seconds.
// build: hipcc --offload-arch=gfx950 -O3 tools/mfma_chain_power_bench.hip -o tools/mfma_chain_power_bench.bin
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#include <vector>
typedef _Float16 half8 __attribute__((ext_vector_type(8)));
typedef float f32x4 __attribute__((ext_vector_type(4)));
constexpr int REPS = 4096, UNROLL = 72;

template <int CH, int RUN, int AF>
__global__ __launch_bounds__(256, 1) void burn(const half8* in, float* out)
{
    half8 a[4], b[4];
#pragma unroll
    for (int i = 0; i < 4; ++i) {
        a[i] = in[(2 * i) * 512 + threadIdx.x % 512];
        b[i] = in[(2 * i + 1) * 512 + threadIdx.x % 512];
    }
    f32x4 acc[CH];
#pragma unroll
    for (int c = 0; c < CH; ++c) acc[c] = f32x4{0.f, 0.f, 0.f, 0.f};
    for (int rep = 0; rep < REPS; ++rep) {
#pragma unroll
        for (int i = 0; i < UNROLL; ++i) {
            const int c = (i / RUN) % CH;
            acc[c] = __builtin_amdgcn_mfma_f32_16x16x32_f16(a[(i / AF) & 3], b[i & 3], acc[c], 0, 0, 0);
        }
    }
    float s = 0;
#pragma unroll
    for (int c = 0; c < CH; ++c) s += acc[c][0] + acc[c][3];
    out[blockIdx.x * 256 + threadIdx.x] = s;
}

template <int CH, int RUN, int AF>
static void run(const half8* in, float* out, double seconds)
{
    hipEvent_t e0, e1;
    hipEventCreate(&e0);
    hipEventCreate(&e1);
    const double flop = 256.0 * 4 * REPS * UNROLL * 16384.0;
    double elapsed = 0;
    while (elapsed < seconds) {
        hipEventRecord(e0);
        for (int i = 0; i < 20; ++i) hipLaunchKernelGGL((burn<CH, RUN, AF>), dim3(256), dim3(256), 0, 0, in, out);
        hipEventRecord(e1);
        hipEventSynchronize(e1);
        float ms;
        hipEventElapsedTime(&ms, e0, e1);
        elapsed += ms * 1e-3;
        if (elapsed >= seconds)
            printf("chains %d, run %d, A fixed for %d: %.1f TFLOP/s  (=> %.0f MHz if the pipe never idles)\n", CH, RUN, AF,
                   20 * flop / (ms * 1e-3) * 1e-12, 20 * flop / (ms * 1e-3) / (1024.0 * 4 * 256) * 1e-6);
    }
}

int main(int argc, char** argv)
{
    const double seconds = argc > 1 ? atof(argv[1]) : 4.0;
    half8* in;
    float* out;
    hipMalloc(&in, 8 * 512 * sizeof(half8));
    hipMalloc(&out, 256 * 512 * sizeof(float));
    std::vector<_Float16> h(8 * 512 * 8);
    srand(1);
    for (auto& v : h) v = (_Float16)(2.f * (rand() / (float)RAND_MAX - 0.5f));
    hipMemcpy(in, h.data(), h.size() * 2, hipMemcpyHostToDevice);
    run<8, 1, 1>(in, out, seconds);
    run<8, 2, 1>(in, out, seconds);
    run<8, 3, 1>(in, out, seconds);
    run<8, 6, 1>(in, out, seconds);
    run<8, 9, 1>(in, out, seconds);
    run<8, 18, 1>(in, out, seconds);
    run<4, 18, 1>(in, out, seconds);
    run<1, 1, 1>(in, out, seconds);
    run<8, 2, 2>(in, out, seconds);
    run<8, 3, 3>(in, out, seconds);
    return 0;
}
