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

template <int WAVES>
__global__ __launch_bounds__(WAVES * 64, 1) void burn(const half8* in, float* out)
{
    half8 a[4], b[4];
#pragma unroll
    for (int i = 0; i < 4; ++i) {
        a[i] = in[(2 * i) * 512 + threadIdx.x % 512];
        b[i] = in[(2 * i + 1) * 512 + threadIdx.x % 512];
    }
    f32x16 acc[2];
#pragma unroll
    for (int c = 0; c < 2; ++c)
#pragma unroll
        for (int r = 0; r < 16; ++r) acc[c][r] = 0.f;
    for (int rep = 0; rep < REPS; ++rep) {
#pragma unroll
        for (int i = 0; i < UNROLL; ++i)
            acc[i & 1] = __builtin_amdgcn_mfma_f32_32x32x16_f16(a[i & 3], b[(i >> 1) & 3], acc[i & 1], 0, 0, 0);
    }
    float s = 0;
#pragma unroll
    for (int c = 0; c < 2; ++c) s += acc[c][0] + acc[c][15];
    out[blockIdx.x * WAVES * 64 + threadIdx.x] = s;
}

template <int WAVES>
static void run(const char* name, const half8* in, float* out, double seconds)
{
    hipEvent_t e0, e1;
    hipEventCreate(&e0);
    hipEventCreate(&e1);
    const double flop = 256.0 * WAVES * REPS * UNROLL * 32768.0;
    double elapsed = 0;
    int n = 0;
    while (elapsed < seconds) {
        hipEventRecord(e0);
        for (int i = 0; i < 20; ++i) hipLaunchKernelGGL(burn<WAVES>, dim3(256), dim3(WAVES * 64), 0, 0, in, out);
        hipEventRecord(e1);
        hipEventSynchronize(e1);
        float ms;
        hipEventElapsedTime(&ms, e0, e1);
        elapsed += ms * 1e-3;
        if (elapsed >= seconds)
            printf("%s t=%5.2fs  %.1f TFLOP/s  (=> %.0f MHz if the pipe never idles)\n", name, elapsed,
                   20 * flop / (ms * 1e-3) * 1e-12, 20 * flop / (ms * 1e-3) / (1024.0 * 4 * 256) * 1e-6);
        ++n;
    }
}

int main(int argc, char** argv)
{
    const double seconds = argc > 1 ? atof(argv[1]) : 8.0;
    half8* in;
    float* out;
    hipMalloc(&in, 8 * 512 * sizeof(half8));
    hipMalloc(&out, 256 * 512 * sizeof(float));
    for (int mode = 0; mode < 2; ++mode) {
        std::vector<_Float16> h(8 * 512 * 8);
        srand(1);
        for (auto& v : h) v = mode ? (_Float16)((rand() / (float)RAND_MAX - 0.5f) * 2.f) : (_Float16)0.f;
        hipMemcpy(in, h.data(), h.size() * 2, hipMemcpyHostToDevice);
        run<4>(mode ? "random, 1 wave/SIMD " : "zeros,  1 wave/SIMD ", in, out, seconds);
        run<8>(mode ? "random, 2 waves/SIMD" : "zeros,  2 waves/SIMD", in, out, seconds);
    }
    return 0;
}
