// Microbenchmark: cycles per ds_read_b128 for a given lane -> LDS byte offset pattern (1 or 4 waves
// per CU, all CUs).  Used to pick the LDS layout of the trunk kernel's B fragments.
// build: hipcc --offload-arch=gfx950 -O3 tools/lds_pattern_bench.hip -o /tmp/lds_pattern_bench
#include <hip/hip_runtime.h>
#include <cstdio>
#include <vector>
#include <functional>
typedef unsigned u32x4 __attribute__((ext_vector_type(4)));

template <int WIDTH>
__global__ __launch_bounds__(512, 1) void k(const int* offs, unsigned* out, long long* cyc, int waves_active)
{
    extern __shared__ __attribute__((aligned(16))) char smem[];
    for (int i = threadIdx.x; i < 40000 / 4; i += 512) ((unsigned*)smem)[i] = i;
    __syncthreads();
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    if (wave >= waves_active) return;
    const char* p = smem + offs[lane];   // re-laundered every repetition below
    u32x4 acc = {0, 0, 0, 0};
    long long t0 = __builtin_amdgcn_s_memtime();
    for (int rep = 0; rep < 256; ++rep) {
        asm volatile("" : "+v"(p));      // not loop-invariant as far as the compiler knows
#pragma unroll
        for (int j = 0; j < 16; ++j) {
            if constexpr (WIDTH == 16) {
                const u32x4 v = *(const u32x4*)(p + j * 256);     // multiples of the bank period: same bank pattern
                acc ^= v;
            } else {
                const uint2 v = *(const uint2*)(p + j * 256);
                acc[0] ^= v.x; acc[1] ^= v.y;
            }
        }
    }
    long long t1 = __builtin_amdgcn_s_memtime();
    out[blockIdx.x * 512 + threadIdx.x] = acc[0] ^ acc[1] ^ acc[2] ^ acc[3];
    if (threadIdx.x == 0 && blockIdx.x == 0) *cyc = t1 - t0;
}

int main()
{
    int* d_offs; unsigned* d_out; long long* d_cyc;
    hipMalloc(&d_offs, 64 * 4); hipMalloc(&d_out, 256 * 512 * 4); hipMalloc(&d_cyc, 8);
    struct Pat { const char* name; std::function<int(int)> f; };
    std::vector<Pat> pats = {
        {"linear: lane*16", [](int l) { return l * 16; }},
        {"32 px x 2 oct, stride 144 (32x32x16 layout)", [](int l) { return (l & 31) * 144 + (l >> 5) * 16; }},
        {"16 px x 4 oct, stride 144 (16x16x32 layout)", [](int l) { return (l & 15) * 144 + (l >> 4) * 16; }},
        {"16 px x 4 oct, stride 128 (no pad)", [](int l) { return (l & 15) * 128 + (l >> 4) * 16; }},
        {"16 px x 4 oct, stride 160", [](int l) { return (l & 15) * 160 + (l >> 4) * 16; }},
        {"16 px x 4 oct, stride 192", [](int l) { return (l & 15) * 192 + (l >> 4) * 16; }},
        {"16 px x 4 oct, stride 208", [](int l) { return (l & 15) * 208 + (l >> 4) * 16; }},
        {"16 px x 4 oct, stride 272", [](int l) { return (l & 15) * 272 + (l >> 4) * 16; }},
        {"16 px x 4 oct, stride 64 (4 oct contiguous per px)", [](int l) { return (l & 15) * 64 + (l >> 4) * 16; }},
        {"16 px x 4 oct, stride 80", [](int l) { return (l & 15) * 80 + (l >> 4) * 16; }},
        {"16 px x 4 oct, oct planes of 16*16+16 B", [](int l) { return (l & 15) * 16 + (l >> 4) * 272; }},
        {"16 px x 4 oct, oct planes contiguous (= linear)", [](int l) { return (l & 15) * 16 + (l >> 4) * 256; }},
        {"16 px x 4 oct, stride 144, oct stride 32", [](int l) { return (l & 15) * 144 + (l >> 4) * 32; }},
        {"16 px x 4 oct, stride 144, oct stride 64", [](int l) { return (l & 15) * 144 + (l >> 4) * 64; }},
    };
    for (int width = 16; width >= 8; width -= 8)
        for (int waves = 1; waves <= 8; waves *= 2)
            for (auto& pt : pats) {
                int h[64];
                for (int l = 0; l < 64; ++l) h[l] = pt.f(l);
                hipMemcpy(d_offs, h, sizeof h, hipMemcpyHostToDevice);
                for (int r = 0; r < 2; ++r) {
                    if (width == 16) hipLaunchKernelGGL(k<16>, dim3(256), dim3(512), 48 * 1024, 0, d_offs, d_out, d_cyc, waves);
                    else hipLaunchKernelGGL(k<8>, dim3(256), dim3(512), 48 * 1024, 0, d_offs, d_out, d_cyc, waves);
                }
                long long c;
                hipMemcpy(&c, d_cyc, 8, hipMemcpyDeviceToHost);
                const double per = c / (256.0 * 16);
                printf("b%-3d %d wave(s)/CU  %-52s %6.1f cycles/read/wave = %5.1f B/clk/CU\n", width * 8, waves, pt.name, per,
                       waves * 64.0 * width / per);
            }
    return 0;
}
