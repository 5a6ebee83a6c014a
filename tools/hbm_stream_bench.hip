// What can a kernel shaped like head_kernel (writes 273 MB of fp16 NHWC activations, reads almost nothing) or like
// tail_kernel (reads those 273 MB, writes 25 MB of bytes) reach on this chip?  The floors for VERDICT r3 item 8.
//   W   write only, 16 B per lane, grid-stride over the array (persistent: 256 x NWG workgroups) or one block per 32 KB
//   R   read only, global_load_dwordx4, sum kept alive
//   D   read only through LDS-DMA (global_load_lds_dwordx4, 1 KB per wave and instruction) into a ring, nothing read back
//   S   strip walk: every workgroup walks DOWN a 34-pixel-wide strip of a 970-wide plane, 4 rows x 34 x 128 B per step
//       by LDS-DMA (trunkw_kernel's access pattern: 4352-byte runs, pitch 124 KB)
// Prints us per pass and TB/s for an array of 2 134 000 pixels x 128 B (the 1080p frame's four reference tiles).
// build: hipcc --offload-arch=gfx950 -O3 tools/hbm_stream_bench.hip -o /tmp/hsb && /tmp/hsb
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#include <vector>
#include <algorithm>

#define CK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { printf("%s: %s\n", #x, hipGetErrorString(e_)); exit(1); } } while (0)

constexpr size_t NPIX = 2134000;
constexpr size_t BYTES = NPIX * 128;

__global__ __launch_bounds__(256) void k_write_tile(uint4* dst, size_t n16)
{
    // one block per 32 KB: 2048 units of 16 B, 8 per thread, each wave writing 1 KB runs
    const size_t base = (size_t)blockIdx.x * 2048;
    const uint4 v = make_uint4(threadIdx.x, blockIdx.x, 3, 4);
#pragma unroll
    for (int i = 0; i < 8; ++i) {
        const size_t q = base + i * 256 + threadIdx.x;
        if (q < n16) dst[q] = v;
    }
}

// head_kernel's store pattern: one block per 8-row x 32-pixel tile of a 970-wide plane (pitch 972 pixels): wave w writes rows
// 2w, 2w+1 of the tile, 4 KB each, 124 KB apart
__global__ __launch_bounds__(256) void k_write_tiles(char* dst, int ntx, int pitch_px)
{
    const int ty = blockIdx.x / ntx, tx = blockIdx.x - ty * ntx;
    const int wave = threadIdx.x >> 6, lane = threadIdx.x & 63;
    const uint4 v = make_uint4(threadIdx.x, blockIdx.x, 3, 4);
#pragma unroll
    for (int n = 0; n < 2; ++n) {
        char* row = dst + ((size_t)(ty * 8 + 2 * wave + n + 1) * pitch_px + tx * 32 + 1) * 128;
#pragma unroll
        for (int k = 0; k < 4; ++k) *(uint4*)(row + (k * 64 + lane) * 16) = v;
    }
}

template <int NT>
__global__ __launch_bounds__(256) void k_write_persist(uint4* dst, size_t n16)
{
    const uint4 v = make_uint4(threadIdx.x, blockIdx.x, 3, 4);
    const size_t stride = (size_t)gridDim.x * 256;
    for (size_t q = (size_t)blockIdx.x * 256 + threadIdx.x; q < n16; q += stride) {
        if constexpr (NT) {
            typedef unsigned u32x4 __attribute__((ext_vector_type(4)));
            const u32x4 vv = {v.x, v.y, v.z, v.w};
            __builtin_nontemporal_store(vv, (u32x4*)(dst + q));
        } else dst[q] = v;
    }
}

__global__ __launch_bounds__(256) void k_read(const uint4* src, size_t n16, unsigned* out)
{
    unsigned acc = 0;
    const size_t stride = (size_t)gridDim.x * 256;
    size_t q = (size_t)blockIdx.x * 256 + threadIdx.x;
    for (; q + 3 * stride < n16; q += 4 * stride) {
        const uint4 a = src[q], b = src[q + stride], c = src[q + 2 * stride], d = src[q + 3 * stride];
        acc += a.x ^ b.y ^ c.z ^ d.w;
    }
    for (; q < n16; q += stride) acc += src[q].x;
    if (acc == 0x12345678u) out[0] = acc;
}

__device__ __forceinline__ void glds16(const void* sbase, unsigned voff, unsigned lds_dst)
{
    unsigned keep;
    asm volatile(
        "s_mov_b32 %0, m0\n\t"
        "s_mov_b32 m0, %3\n\t"
        "s_nop 0\n\t"
        "global_load_lds_dwordx4 %1, %2\n\t"
        "s_mov_b32 m0, %0"
        : "=&s"(keep)
        : "v"(voff), "s"(sbase), "s"(lds_dst)
        : "memory");
}

// linear LDS-DMA stream: each wave fetches 1 KB pieces round-robin into its own 16 KB ring
template <int WAVES>
__global__ __launch_bounds__(64 * WAVES) void k_dma(const char* src, size_t bytes)
{
    extern __shared__ __attribute__((aligned(16))) char lds[];
    const int wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6), lane = threadIdx.x & 63;
    const unsigned lds0 = (unsigned)(size_t)lds + wave * 16384;
    const size_t npieces = bytes / 1024;
    const size_t stride = (size_t)gridDim.x * WAVES;
    int slot = 0;
    for (size_t pc = (size_t)blockIdx.x * WAVES + wave; pc < npieces; pc += stride) {
        const char* base = src + (pc << 10);
        glds16(base, lane * 16, lds0 + slot * 1024);
        slot = (slot + 1) & 15;
        asm volatile("s_waitcnt vmcnt(12)" ::: "memory");
    }
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
}

// strip walk: planes of W = 970 columns (pitch 972 pixels), H rows; strips of 30 columns + 4 halo; each workgroup
// (8 waves) takes strips round-robin and walks them in 4-row steps: 17 pieces of 1 KB per step (4 rows x 4352 B),
// piece c = units [64c, 64c+64) of the step, unit q -> row q / 272, byte (q % 272) * 16
__global__ __launch_bounds__(512) void k_strip(const char* src, int H, int W, int pitch_px, int nstrips_total, int ring_steps, int nseg)
{
    extern __shared__ __attribute__((aligned(16))) char lds[];
    const int wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6), lane = threadIdx.x & 63;
    const unsigned lds0 = (unsigned)(size_t)lds;
    const int strips_per_plane = (W + 29) / 30;
    int slot = 0;
    for (int item = blockIdx.x; item < nstrips_total * nseg; item += gridDim.x) {
        const int s = item / nseg, sg = item - s * nseg;
        const int plane = s / strips_per_plane, k = s - plane * strips_per_plane;
        const char* pbase = src + ((size_t)plane * (H + 2) * pitch_px + (size_t)k * 30) * 128;
        const int rows = H / nseg;
        for (int y = sg * rows; y + 4 <= (sg + 1) * rows; y += 4) {
            const char* base = pbase + (size_t)y * pitch_px * 128;
            for (int c = wave; c < 17; c += 8) {
                const int q = c * 64 + lane;
                const int r = q / 272, u = q - r * 272;
                glds16(base, (unsigned)(r * pitch_px * 128 + u * 16), lds0 + slot * 17408 + c * 1024);
            }
            slot = slot + 1 == ring_steps ? 0 : slot + 1;
            asm volatile("s_waitcnt vmcnt(6)" ::: "memory");
        }
    }
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
}

template <typename F>
static void timeit(const char* name, double bytes, F&& launch)
{
    hipEvent_t e0, e1;
    CK(hipEventCreate(&e0));
    CK(hipEventCreate(&e1));
    for (int i = 0; i < 5; ++i) launch();
    CK(hipDeviceSynchronize());
    std::vector<float> ts;
    for (int rep = 0; rep < 5; ++rep) {
        CK(hipEventRecord(e0));
        for (int i = 0; i < 20; ++i) launch();
        CK(hipEventRecord(e1));
        CK(hipEventSynchronize(e1));
        float ms;
        CK(hipEventElapsedTime(&ms, e0, e1));
        ts.push_back(ms / 20);
    }
    std::sort(ts.begin(), ts.end());
    printf("%-58s %7.1f us   %5.2f TB/s\n", name, ts[2] * 1e3, bytes / (ts[2] * 1e-3) / 1e12);
    CK(hipGetLastError());
}

int main()
{
    // NBUF arrays per direction, used in turn: a pass over the SAME 273 MB again and again is served in part by the 256 MB
    // Infinity Cache (first version of this file: write-only 39 us = 7.0 TB/s, which no HBM3E stack delivers)
    constexpr int NBUF = 6;
    char *abuf[NBUF], *bbuf[NBUF];
    unsigned* out;
    for (int i = 0; i < NBUF; ++i) {
        CK(hipMalloc(&abuf[i], BYTES + (1 << 20)));
        CK(hipMalloc(&bbuf[i], BYTES + (1 << 20)));
        CK(hipMemset(abuf[i], 1, BYTES));
        CK(hipMemset(bbuf[i], 2, BYTES));
    }
    CK(hipMalloc(&out, 64));
    int turn = 0;
#define a (abuf[(turn++) % NBUF])
#define b (bbuf[(turn++) % NBUF])
    const size_t n16 = BYTES / 16;
    printf("array: %.1f MB, %d arrays per direction used in turn\n", BYTES / 1e6, NBUF);
    timeit("W  one block per 32 KB (8336 blocks of 256)", BYTES, [&] { k_write_tile<<<(unsigned)((n16 + 2047) / 2048), 256>>>((uint4*)a, n16); });
    {
        const int ntx = 30, nty = 270;      // 30 full tiles per row of tiles, pitch 972: 270 x 8 + 2 rows stay inside the array
        timeit("W  head_kernel's pattern: 8 x 32-pixel tiles, rows 124 KB apart", (double)ntx * nty * 32768, [&] { k_write_tiles<<<ntx * nty, 256>>>(a, ntx, 972); });
    }
    for (int nwg : {1, 2, 4, 8}) {
        char nm[96];
        snprintf(nm, sizeof nm, "W  persistent, %d x 256 blocks of 256", nwg);
        timeit(nm, BYTES, [&] { k_write_persist<0><<<256 * nwg, 256>>>((uint4*)a, n16); });
    }
    timeit("W  persistent, 4 x 256 blocks, nontemporal stores", BYTES, [&] { k_write_persist<1><<<1024, 256>>>((uint4*)a, n16); });
    for (int nwg : {2, 4, 8}) {
        char nm[96];
        snprintf(nm, sizeof nm, "R  global_load_dwordx4, %d x 256 blocks of 256", nwg);
        timeit(nm, BYTES, [&] { k_read<<<256 * nwg, 256>>>((const uint4*)b, n16, out); });
    }
    {
        CK(hipFuncSetAttribute((const void*)k_dma<4>, hipFuncAttributeMaxDynamicSharedMemorySize, 65536));
        CK(hipFuncSetAttribute((const void*)k_dma<8>, hipFuncAttributeMaxDynamicSharedMemorySize, 131072));
        timeit("D  LDS-DMA 1 KB pieces, 256 blocks x 8 waves, 13 in flight", BYTES, [&] { k_dma<8><<<256, 512, 131072>>>(b, BYTES); });
        timeit("D  LDS-DMA 1 KB pieces, 512 blocks x 4 waves, 13 in flight", BYTES, [&] { k_dma<4><<<512, 256, 65536>>>(b, BYTES); });
    }
    {
        // the 1080p frame's planes: 970 x 970 x 2 and 130 x 970 x 2 -> as 2200 rows of one plane class: use H = 1100 x 2 planes
        const int W = 970, pitch = 972, H = 1096, planes = 2;
        const int nstr = planes * ((W + 29) / 30);
        CK(hipFuncSetAttribute((const void*)k_strip, hipFuncAttributeMaxDynamicSharedMemorySize, 4 * 17408));
        const double useful = (double)planes * H * W * 128;
        CK(hipFuncSetAttribute((const void*)k_strip, hipFuncAttributeMaxDynamicSharedMemorySize, 8 * 17408));
        timeit("S  strip walk (34 columns fetched per 30), 66 strips x 4 segments, ring 4 steps", useful, [&] { k_strip<<<nstr * 4, 512, 4 * 17408>>>(b, H, W, pitch, nstr, 4, 4); });
        timeit("S  the same, ring 8 steps (136 KB in flight per CU)", useful, [&] { k_strip<<<nstr * 4, 512, 8 * 17408>>>(b, H, W, pitch, nstr, 8, 4); });
        timeit("S  66 strips x 8 segments, ring 4 steps, two workgroups per CU", useful, [&] { k_strip<<<nstr * 8, 512, 4 * 17408>>>(b, H, W, pitch, nstr, 4, 8); });
    }
    return 0;
}
