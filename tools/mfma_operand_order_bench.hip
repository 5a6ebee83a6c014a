// Does the ORDER in which a k-loop visits its operands matter at the power cap?  All-chip sustained loops of
// v_mfma_f32_16x16x32_f16 on register operands (random fp16 data), one wave per SIMD, 16 accumulators like trunkw_kernel's,
// six "fragments" (B) and twelve "weights" (A) in registers; only the visiting order differs:
//   0  every MFMA changes A and B (no reuse between neighbours)
//   1  fragment-major, trunkw_kernel's order: B fixed over the up-to-three MFMAs (tap rows) of a fragment, A changes
//   2  weight-major: A fixed over four consecutive MFMAs (output rows), B changes
//   3  pairs: A fixed over two consecutive MFMAs (two neighbouring rows' fragments), B alternates
// build: hipcc --offload-arch=gfx950 -O3 tools/mfma_operand_order_bench.hip -o /tmp/mfma_order ; run: /tmp/mfma_order [seconds=6]
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#include <vector>
typedef _Float16 half8 __attribute__((ext_vector_type(8)));
typedef float f32x4 __attribute__((ext_vector_type(4)));
constexpr int REPS = 4096;

template <int ORDER>
__global__ __launch_bounds__(256, 1) void burn(const half8* in, float* out)
{
    half8 w[12], b[6];
#pragma unroll
    for (int i = 0; i < 12; ++i) w[i] = in[i * 256 + threadIdx.x];
#pragma unroll
    for (int i = 0; i < 6; ++i) b[i] = in[(12 + i) * 256 + threadIdx.x];
    f32x4 acc[4][4];
#pragma unroll
    for (int n = 0; n < 4; ++n)
#pragma unroll
        for (int j = 0; j < 4; ++j) acc[n][j] = f32x4{0.f, 0.f, 0.f, 0.f};
    for (int rep = 0; rep < REPS; ++rep) {
        // one "(channel half, j)" group of a k-loop: 6 fragments (rows R), 12 MFMAs: acc[n][j] += w[dy] * b[n + dy]
#pragma unroll
        for (int j = 0; j < 4; ++j) {
            if constexpr (ORDER == 0) {
#pragma unroll
                for (int k = 0; k < 12; ++k) { const int n = k & 3, dy = (k >> 2); acc[n][j] = __builtin_amdgcn_mfma_f32_16x16x32_f16(w[j * 3 + ((dy + k) % 3)], b[(n + dy + k) % 6], acc[n][j], 0, 0, 0); }
            } else if constexpr (ORDER == 1) {
#pragma unroll
                for (int R = 0; R < 6; ++R)
#pragma unroll
                    for (int n = 0; n < 4; ++n) { const int dy = R - n; if (dy >= 0 && dy <= 2) acc[n][j] = __builtin_amdgcn_mfma_f32_16x16x32_f16(w[j * 3 + dy], b[R], acc[n][j], 0, 0, 0); }
            } else if constexpr (ORDER == 2) {
#pragma unroll
                for (int dy = 0; dy < 3; ++dy)
#pragma unroll
                    for (int n = 0; n < 4; ++n) acc[n][j] = __builtin_amdgcn_mfma_f32_16x16x32_f16(w[j * 3 + dy], b[n + dy], acc[n][j], 0, 0, 0);
            } else {
#pragma unroll
                for (int Rp = 0; Rp < 6; Rp += 2)
#pragma unroll
                    for (int dy = 0; dy < 3; ++dy)
#pragma unroll
                        for (int q = 0; q < 2; ++q) { const int n = Rp + q - dy; if (n >= 0 && n < 4) acc[n][j] = __builtin_amdgcn_mfma_f32_16x16x32_f16(w[j * 3 + dy], b[Rp + q], acc[n][j], 0, 0, 0); }
            }
        }
    }
    float s = 0;
#pragma unroll
    for (int n = 0; n < 4; ++n)
#pragma unroll
        for (int j = 0; j < 4; ++j) s += acc[n][j][0] + acc[n][j][3];
    out[blockIdx.x * 256 + threadIdx.x] = s;
}

template <int ORDER>
static void run(const char* name, const half8* in, float* out, double seconds)
{
    hipEvent_t e0, e1;
    hipEventCreate(&e0);
    hipEventCreate(&e1);
    const double flop = 256.0 * 4 * REPS * 48 * 16384.0;
    double elapsed = 0;
    while (elapsed < seconds) {
        hipEventRecord(e0);
        for (int i = 0; i < 20; ++i) hipLaunchKernelGGL((burn<ORDER>), dim3(256), dim3(256), 0, 0, in, out);
        hipEventRecord(e1);
        hipEventSynchronize(e1);
        float ms;
        hipEventElapsedTime(&ms, e0, e1);
        elapsed += ms * 1e-3;
        if (elapsed >= seconds) printf("%-78s %.1f TFLOP/s\n", name, 20 * flop / (ms * 1e-3) * 1e-12);
    }
}

int main(int argc, char** argv)
{
    const double seconds = argc > 1 ? atof(argv[1]) : 6.0;
    half8* in;
    float* out;
    hipMalloc(&in, 18 * 256 * sizeof(half8));
    hipMalloc(&out, 256 * 256 * sizeof(float));
    std::vector<_Float16> h(18 * 256 * 8);
    srand(1);
    for (auto& v : h) v = (_Float16)(2.f * (rand() / (float)RAND_MAX - 0.5f));
    hipMemcpy(in, h.data(), h.size() * 2, hipMemcpyHostToDevice);
    for (int round = 0; round < 2; ++round) {
        run<0>("0 every MFMA changes A and B", in, out, seconds);
        run<1>("1 fragment-major (trunkw_kernel: B fixed over a fragment's tap rows)", in, out, seconds);
        run<2>("2 weight-major (A fixed over four output rows)", in, out, seconds);
        run<3>("3 pairs (A fixed over two neighbouring rows' fragments)", in, out, seconds);
    }
    return 0;
}
