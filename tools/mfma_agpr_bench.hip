// Does it matter for power where the MFMA accumulators live?  Same sustained all-chip loop as
// mfma_power_bench.hip (v_mfma_f32_16x16x32_f16, random fp16 operands), accumulators pinned to
// ArchVGPRs or to AccVGPRs by inline asm.
// build: hipcc --offload-arch=gfx950 -O3 tools/mfma_agpr_bench.hip -o /tmp/mfma_agpr_bench
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#include <vector>
typedef _Float16 half8 __attribute__((ext_vector_type(8)));
typedef float f32x4 __attribute__((ext_vector_type(4)));
constexpr int REPS = 8192, UNROLL = 16;

template <int AGPR>
__global__ __launch_bounds__(256, 1) void burn(const half8* in, float* out)
{
    half8 a[4], b[4];
#pragma unroll
    for (int i = 0; i < 4; ++i) { a[i] = in[(2 * i) * 256 + threadIdx.x]; b[i] = in[(2 * i + 1) * 256 + threadIdx.x]; }
    f32x4 acc[4];
#pragma unroll
    for (int c = 0; c < 4; ++c) acc[c] = f32x4{0.f, 0.f, 0.f, 0.f};
    for (int rep = 0; rep < REPS; ++rep) {
#pragma unroll
        for (int i = 0; i < UNROLL; ++i) {
            if constexpr (AGPR)
                asm volatile("v_mfma_f32_16x16x32_f16 %0, %1, %2, %0" : "+a"(acc[i & 3]) : "v"(a[i & 3]), "v"(b[(i >> 2) & 3]));
            else
                asm volatile("v_mfma_f32_16x16x32_f16 %0, %1, %2, %0" : "+v"(acc[i & 3]) : "v"(a[i & 3]), "v"(b[(i >> 2) & 3]));
        }
    }
    float s = 0;
#pragma unroll
    for (int c = 0; c < 4; ++c) s += acc[c][0] + acc[c][3];
    out[blockIdx.x * 256 + threadIdx.x] = s;
}

template <int AGPR>
static void run(const char* name, const half8* in, float* out, double seconds)
{
    hipEvent_t e0, e1;
    hipEventCreate(&e0); hipEventCreate(&e1);
    const double flop = 256.0 * 4 * REPS * UNROLL * 16384.0;
    double elapsed = 0;
    while (elapsed < seconds) {
        hipEventRecord(e0);
        for (int i = 0; i < 20; ++i) hipLaunchKernelGGL(burn<AGPR>, dim3(256), dim3(256), 0, 0, in, out);
        hipEventRecord(e1);
        hipEventSynchronize(e1);
        float ms; hipEventElapsedTime(&ms, e0, e1);
        elapsed += ms * 1e-3;
        if (elapsed >= seconds) printf("%-40s %.1f TFLOP/s\n", name, 20 * flop / (ms * 1e-3) * 1e-12);
    }
}

int main(int argc, char** argv)
{
    const double seconds = argc > 1 ? atof(argv[1]) : 4.0;
    half8* in; float* out;
    hipMalloc(&in, 8 * 256 * sizeof(half8)); hipMalloc(&out, 256 * 256 * sizeof(float));
    std::vector<_Float16> h(8 * 256 * 8);
    srand(1);
    for (auto& v : h) v = (_Float16)((rand() / (float)RAND_MAX - 0.5f) * 2.f);
    hipMemcpy(in, h.data(), h.size() * 2, hipMemcpyHostToDevice);
    for (int r = 0; r < 2; ++r) {
        run<0>("accumulators in ArchVGPRs", in, out, seconds);
        run<1>("accumulators in AccVGPRs", in, out, seconds);
    }
    return 0;
}
