// Microbenchmark: cycles per ring-store of sub10_kernel's epilogue for candidate layouts of a 24-channel fp16 pixel (48 bytes).
// A lane (o, p) of an MFMA result owns channels 4o..4o+3 (8 bytes) and 16+2o, 17+2o (4 bytes) of pixel pix(p); a fragment is 16
// pixels.  Per fragment and lane: ONE 8-byte and ONE 4-byte store (or one 12-byte store).  8 waves per CU store at once, like the
// kernel's eight trunk waves.  (tools/experiments/sub10_linear_stores.patch measured the kernel 3.2 % faster with lane-linear
// stores: which REAL layout gets near that?)
// build: hipcc --offload-arch=gfx950 -O3 tools/lds_store_pattern_bench.hip -o /tmp/lds_store_bench
#include <hip/hip_runtime.h>
#include <cstdio>
#include <functional>
#include <vector>

typedef unsigned u32x2 __attribute__((ext_vector_type(2)));
typedef unsigned u32x3 __attribute__((ext_vector_type(3)));

template <int MODE>   // 0: b64 + b32, 1: one 12-byte store
__global__ __launch_bounds__(512, 1) void k(const int* off8, const int* off4, unsigned* out, long long* cyc)
{
    extern __shared__ __attribute__((aligned(16))) char smem[];
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    unsigned p8 = (unsigned)(size_t)(const __attribute__((address_space(3))) char*)smem + wave * 8192 + off8[lane];
    unsigned p4 = (unsigned)(size_t)(const __attribute__((address_space(3))) char*)smem + wave * 8192 + off4[lane];
    unsigned v = threadIdx.x;
    u32x2 v2 = {v, v + 1};
    u32x3 v3 = {v, v + 1, v + 2};
    __syncthreads();
    long long t0 = __builtin_amdgcn_s_memtime();
    for (int rep = 0; rep < 256; ++rep) {
        // five fragments of a row, 768 bytes apart (explicit instructions: hipcc would drop stores nobody reads)
        if constexpr (MODE == 0) {
            asm volatile("ds_write_b64 %0, %1\n\tds_write_b32 %2, %3\n\t"
                         "ds_write_b64 %0, %1 offset:768\n\tds_write_b32 %2, %3 offset:768\n\t"
                         "ds_write_b64 %0, %1 offset:1536\n\tds_write_b32 %2, %3 offset:1536\n\t"
                         "ds_write_b64 %0, %1 offset:2304\n\tds_write_b32 %2, %3 offset:2304\n\t"
                         "ds_write_b64 %0, %1 offset:3072\n\tds_write_b32 %2, %3 offset:3072" ::"v"(p8), "v"(v2), "v"(p4), "v"(v) : "memory");
        } else {
            asm volatile("ds_write_b96 %0, %1\n\tds_write_b96 %0, %1 offset:768\n\tds_write_b96 %0, %1 offset:1536\n\t"
                         "ds_write_b96 %0, %1 offset:2304\n\tds_write_b96 %0, %1 offset:3072" ::"v"(p8), "v"(v3) : "memory");
        }
    }
    asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
    long long t1 = __builtin_amdgcn_s_memtime();
    out[blockIdx.x * 512 + threadIdx.x] = ((unsigned*)smem)[threadIdx.x];
    if (threadIdx.x == 0 && blockIdx.x == 0) *cyc = t1 - t0;
}

static int pix(int p) { return p < 4 ? 2 * p : p >= 12 ? 2 * (p - 8) : 2 * (p - 4) + 1; }

int main()
{
    int *d8, *d4; unsigned* d_out; long long* d_cyc;
    hipMalloc(&d8, 256); hipMalloc(&d4, 256); hipMalloc(&d_out, 256 * 512 * 4); hipMalloc(&d_cyc, 8);
    struct Pat { const char* name; int mode; std::function<int(int, int)> f8, f4; };     // (o, p) -> byte offset
    std::vector<Pat> pats = {
        {"shipped: 8 B at pix*48 + 8o, 4 B at pix*48 + 32 + 4o", 0, [](int o, int p) { return pix(p) * 48 + 8 * o; }, [](int o, int p) { return pix(p) * 48 + 32 + 4 * o; }},
        {"lane-linear (the timing experiment: not a layout)", 0, [](int o, int p) { return (o * 16 + p) * 8; }, [](int o, int p) { return 512 + (o * 16 + p) * 4; }},
        {"12 B slots: 8 B at pix*48 + 12o (+4 if o odd), 4 B beside it (all aligned)", 0, [](int o, int p) { return pix(p) * 48 + 12 * o + (o & 1 ? 4 : 0); }, [](int o, int p) { return pix(p) * 48 + 12 * o + (o & 1 ? 0 : 8); }},
        {"12 B slots: ONE 12-byte store at pix*48 + 12o", 1, [](int o, int p) { return pix(p) * 48 + 12 * o; }, [](int o, int p) { return 0; }},
        {"shipped offsets, pixels in natural order p (no even/odd permutation)", 0, [](int o, int p) { return p * 48 + 8 * o; }, [](int o, int p) { return p * 48 + 32 + 4 * o; }},
        {"8 B planes: 8 B at o*128 + pix*8, 4 B at 512 + o*64 + pix*4 (channel-plane layout)", 0, [](int o, int p) { return o * 128 + pix(p) * 8; }, [](int o, int p) { return 512 + o * 64 + pix(p) * 4; }},
    };
    for (auto& pt : pats) {
        int h8[64], h4[64];
        for (int l = 0; l < 64; ++l) { h8[l] = pt.f8(l >> 4, l & 15); h4[l] = pt.f4(l >> 4, l & 15); }
        hipMemcpy(d8, h8, sizeof h8, hipMemcpyHostToDevice);
        hipMemcpy(d4, h4, sizeof h4, hipMemcpyHostToDevice);
        long long c = 0;
        for (int r = 0; r < 3; ++r) {
            if (pt.mode == 0) hipLaunchKernelGGL(k<0>, dim3(256), dim3(512), 66 * 1024, 0, d8, d4, d_out, d_cyc);
            else hipLaunchKernelGGL(k<1>, dim3(256), dim3(512), 66 * 1024, 0, d8, d4, d_out, d_cyc);
            hipMemcpy(&c, d_cyc, 8, hipMemcpyDeviceToHost);
        }
        printf("%-88s %6.1f cycles per fragment and wave (8 waves storing)\n", pt.name, c / (256.0 * 5));
    }
    return 0;
}
