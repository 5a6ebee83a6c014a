// Micro-benchmark (gfx950): how many instructions of kind T does a wave get through while the OTHER wave of its SIMD
// issues v_mfma_f32_16x16x32_f16 back to back?  Workgroup of 8 waves: waves 0..3 (one per SIMD) run the MFMA loop,
// waves 4..7 (the same SIMDs) run instruction kind T until the MFMA wave of their SIMD raises a flag in LDS.
//   hipcc --offload-arch=gfx950 -O3 -o /tmp/coissue tools/coissue_bench.hip && /tmp/coissue
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#include <vector>
typedef _Float16 half8 __attribute__((ext_vector_type(8)));
typedef _Float16 half2v __attribute__((ext_vector_type(2)));
typedef float f32x4 __attribute__((ext_vector_type(4)));
typedef float f32x2 __attribute__((ext_vector_type(2)));

#define MEMTIME() __builtin_amdgcn_s_memtime()

enum { T_NONE, T_ADD32, T_PKADD32, T_PKFMA32, T_PKMUL16, T_PKMAX16, T_CVT, T_DPP, T_DSW64, T_GST64, T_FMA32, T_MIX, T_DSR128, T_MOV, T_PKADD16, T_CVTRNE, T_SWAP, T_DSW128, T_GST128, T_DSW2, NKIND };
static const char* const KIND[NKIND] = {"(nothing)", "v_add_f32", "v_pk_add_f32", "v_pk_fma_f32", "v_pk_mul_f16", "v_pk_max_f16", "v_cvt_pk_f16_f32(rtz)",
                                         "v_mov_b32 dpp row_shl:1", "ds_write_b64", "global_store_dwordx2", "v_fma_f32", "epilogue mix", "ds_read_b128", "v_mov_b32", "v_pk_add_f16", "v_cvt_pk_f16_f32 (RNE)", "v_permlane16_swap_b32", "ds_write_b128", "global_store_dwordx4", "ds_write2_b64"};

template <int T>
__device__ __forceinline__ void block16(f32x2 (&x)[8], const f32x2 c, char* lds, char* gm)
{
    // 16 instructions of kind T on 8 independent chains (dependent distance 8)
#pragma unroll
    for (int r = 0; r < 2; ++r)
#pragma unroll
        for (int i = 0; i < 8; ++i) {
            if constexpr (T == T_ADD32) asm volatile("v_add_f32 %0, %0, %1" : "+v"(x[i][0]) : "v"(c[0]));
            if constexpr (T == T_FMA32) asm volatile("v_fma_f32 %0, %0, %1, %1" : "+v"(x[i][0]) : "v"(c[0]));
            if constexpr (T == T_MOV) asm volatile("v_mov_b32 %0, %1" : "+v"(x[i][0]) : "v"(x[(i + 1) & 7][1]));
            if constexpr (T == T_PKADD32) asm volatile("v_pk_add_f32 %0, %0, %1" : "+v"(x[i]) : "v"(c));
            if constexpr (T == T_PKFMA32) asm volatile("v_pk_fma_f32 %0, %0, %1, %1" : "+v"(x[i]) : "v"(c));
            if constexpr (T == T_PKMUL16) asm volatile("v_pk_mul_f16 %0, %0, %1" : "+v"(x[i][0]) : "v"(c[0]));
            if constexpr (T == T_PKADD16) asm volatile("v_pk_add_f16 %0, %0, %1" : "+v"(x[i][0]) : "v"(c[0]));
            if constexpr (T == T_CVTRNE) asm volatile("v_cvt_pk_f16_f32 %0, %1, %2" : "=v"(x[i][0]) : "v"(x[(i + 1) & 7][1]), "v"(c[0]));
            if constexpr (T == T_SWAP) asm volatile("v_permlane16_swap_b32 %0, %1" : "+v"(x[i][0]), "+v"(x[i][1]));
            if constexpr (T == T_DSW128) { f32x4 t = {x[i][0], x[i][1], x[i][0], x[i][1]}; asm volatile("ds_write_b128 %0, %1" : : "v"((unsigned)(size_t)lds), "v"(t) : "memory"); }
            if constexpr (T == T_DSW2) asm volatile("ds_write2_b64 %0, %1, %2 offset0:0 offset1:128" : : "v"((unsigned)(size_t)lds), "v"(x[i]), "v"(x[(i + 1) & 7]) : "memory");
            if constexpr (T == T_GST128) { f32x4 t = {x[i][0], x[i][1], x[i][0], x[i][1]}; asm volatile("global_store_dwordx4 %0, %1, off" : : "v"(gm), "v"(t) : "memory"); }
            if constexpr (T == T_PKMAX16) asm volatile("v_pk_max_f16 %0, %0, %1" : "+v"(x[i][0]) : "v"(c[0]));
            if constexpr (T == T_CVT) asm volatile("v_cvt_pkrtz_f16_f32 %0, %1, %2" : "=v"(x[i][0]) : "v"(x[(i + 1) & 7][1]), "v"(c[0]));
            if constexpr (T == T_DPP) asm volatile("v_mov_b32_dpp %0, %1 row_shl:1 row_mask:0xf bank_mask:0xf" : "+v"(x[i][0]) : "v"(x[(i + 1) & 7][1]));
            if constexpr (T == T_DSW64) asm volatile("ds_write_b64 %0, %1 offset:%2" : : "v"((unsigned)(size_t)lds), "v"(x[i]), "n"(0) : "memory");
            if constexpr (T == T_DSR128) { f32x4 t; asm volatile("ds_read_b128 %0, %1" : "=v"(t) : "v"((unsigned)(size_t)lds) : "memory"); x[i][0] = t[0]; }
            if constexpr (T == T_GST64) asm volatile("global_store_dwordx2 %0, %1, off" : : "v"(gm), "v"(x[i]) : "memory");
        }
    if constexpr (T == T_DSW64 || T == T_DSW128 || T == T_DSW2) asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
    if constexpr (T == T_GST128) asm volatile("s_waitcnt vmcnt(8)" ::: "memory");
    if constexpr (T == T_DSR128) asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
    if constexpr (T == T_GST64) asm volatile("s_waitcnt vmcnt(8)" ::: "memory");
    if constexpr (T == T_MIX) {
        // the shape of one epilogue row: 8 packed f32, 4 converts, 4 + 4 packed f16, 2 DPP, 2 ds_write_b64  (24 instructions)
#pragma unroll
        for (int i = 0; i < 4; ++i) asm volatile("v_pk_add_f32 %0, %0, %1" : "+v"(x[i]) : "v"(x[i + 4]));
#pragma unroll
        for (int i = 0; i < 4; ++i) asm volatile("v_pk_fma_f32 %0, %0, %1, %2" : "+v"(x[i + 4]) : "v"(c), "v"(x[i]));
#pragma unroll
        for (int i = 0; i < 4; ++i) asm volatile("v_cvt_pkrtz_f16_f32 %0, %1, %2" : "=v"(x[i][0]) : "v"(x[i + 4][0]), "v"(x[i + 4][1]));
#pragma unroll
        for (int i = 0; i < 4; ++i) asm volatile("v_pk_mul_f16 %0, %1, %2" : "=v"(x[i][1]) : "v"(x[i][0]), "v"(c[0]));
#pragma unroll
        for (int i = 0; i < 4; ++i) asm volatile("v_pk_max_f16 %0, %0, %1" : "+v"(x[i][0]) : "v"(x[i][1]));
        asm volatile("v_mov_b32_dpp %0, %1 row_shl:1 row_mask:0xf bank_mask:0xf" : "+v"(x[4][0]) : "v"(x[0][0]));
        asm volatile("v_mov_b32_dpp %0, %1 row_shl:1 row_mask:0xf bank_mask:0xf" : "+v"(x[5][0]) : "v"(x[1][0]));
        asm volatile("ds_write_b64 %0, %1" : : "v"((unsigned)(size_t)lds), "v"(x[0]) : "memory");
        asm volatile("ds_write_b64 %0, %1 offset:2048" : : "v"((unsigned)(size_t)lds), "v"(x[1]) : "memory");
    }
}

struct Out { unsigned long long t_mfma, t_other, n_other; };

// MODE 0: both; 1: the MFMA waves alone (the others exit); 2: the others alone for a fixed number of blocks
template <int T, int LDSREAD>
__global__ __launch_bounds__(512, 1) void k(Out* out, int nmfma16, int mode, int nfixed, char* gm)
{
    extern __shared__ char smem[];
    volatile int* flag = (volatile int*)(smem + 65536);
    const int wave = threadIdx.x >> 6, lane = threadIdx.x & 63, simd = wave & 3;
    if (threadIdx.x < 4) flag[threadIdx.x * 16] = 0;
    __syncthreads();
    char* mylds = smem + (wave * 64 + lane) * 16;
    if (wave < 4) {
        if (mode == 2) return;
        half8 a, b;
        for (int i = 0; i < 8; ++i) { a[i] = (_Float16)(lane * 0.001f + i); b[i] = (_Float16)(1.f - i * 0.01f); }
        f32x4 acc[4] = {{0, 0, 0, 0}, {0, 0, 0, 0}, {0, 0, 0, 0}, {0, 0, 0, 0}};
        const unsigned long long t0 = MEMTIME();
        if constexpr (LDSREAD) {
            // the k-loop's shape: the fragments (one ds_read_b128 each, three MFMAs each) are read four fragments ahead
            half8 cur[4], nxt[4];
            for (int i = 0; i < 4; ++i) cur[i] = *(const half8*)(mylds + i * 1024);
            for (int it = 0; it < nmfma16 * 4 / 3; ++it) {
#pragma unroll
                for (int i = 0; i < 4; ++i) {
                    asm volatile("" : "+v"(cur[i]));
#pragma unroll
                    for (int u = 0; u < 3; ++u) acc[(i * 3 + u) & 3] = __builtin_amdgcn_mfma_f32_16x16x32_f16(a, cur[i], acc[(i * 3 + u) & 3], 0, 0, 0);
                    __builtin_amdgcn_sched_barrier(0);
                    nxt[i] = *(const half8*)(mylds + (((it & 1) * 4 + i) * 1024));
                    __builtin_amdgcn_sched_barrier(0);
                }
#pragma unroll
                for (int i = 0; i < 4; ++i) cur[i] = nxt[i];
            }
        } else {
            for (int it = 0; it < nmfma16; ++it) {
#pragma unroll
                for (int u = 0; u < 16; ++u) acc[u & 3] = __builtin_amdgcn_mfma_f32_16x16x32_f16(a, b, acc[u & 3], 0, 0, 0);
            }
        }
        asm volatile("s_nop 7\n\ts_nop 7" ::: "memory");
        const unsigned long long t1 = MEMTIME();
        flag[simd * 16] = 1;
        float s = 0;
        for (int i = 0; i < 4; ++i) s += acc[i][0] + acc[i][3];
        if (lane == 0) { out[blockIdx.x * 8 + wave].t_mfma = t1 - t0; out[blockIdx.x * 8 + wave].n_other = (unsigned long long)s; }
    } else {
        if (mode == 1) return;
        f32x2 x[8], c = {1.0001f, 0.5f};
        for (int i = 0; i < 8; ++i) x[i] = f32x2{(float)lane + i, (float)i};
        unsigned long long n = 0;
        const unsigned long long t0 = MEMTIME();
        if (mode == 2) {
            for (int it = 0; it < nfixed; ++it) { block16<T>(x, c, mylds, gm + (size_t)(blockIdx.x * 512 + threadIdx.x) * 16); ++n; }
        } else {
            while (flag[simd * 16] == 0) {
#pragma unroll
                for (int r = 0; r < 4; ++r) block16<T>(x, c, mylds, gm + (size_t)(blockIdx.x * 512 + threadIdx.x) * 16);
                n += 4;
            }
        }
        const unsigned long long t1 = MEMTIME();
        float s = 0;
        for (int i = 0; i < 8; ++i) s += x[i][0] + x[i][1];
        if (lane == 0) { out[blockIdx.x * 8 + wave].t_other = t1 - t0; out[blockIdx.x * 8 + wave].n_other = n; out[blockIdx.x * 8 + wave].t_mfma = (unsigned long long)s; }
    }
}

template <int T, int LDSREAD>
void run(Out* d_out, char* gm, int grid)
{
    const int nm16 = 600;  // 9600 MFMAs
    std::vector<Out> h(grid * 8);
    auto launch = [&](int mode, int nfixed) {
        hipMemset(d_out, 0, sizeof(Out) * grid * 8);
        hipLaunchKernelGGL((k<T, LDSREAD>), dim3(grid), dim3(512), 65536 + 1024, 0, d_out, nm16, mode, nfixed, gm);
        hipDeviceSynchronize();
        hipMemcpy(h.data(), d_out, sizeof(Out) * grid * 8, hipMemcpyDeviceToHost);
    };
    hipFuncSetAttribute((const void*)k<T, LDSREAD>, hipFuncAttributeMaxDynamicSharedMemorySize, 65536 + 1024);
    const int per = T == T_MIX ? 24 : 16;
    launch(1, 0); launch(1, 0);
    const double m_alone = (double)h[0].t_mfma / (nm16 * 16);
    launch(2, 2000);
    const double o_alone = (double)h[4].t_other / (2000.0 * per);
    launch(0, 0);
    const double m_both = (double)h[0].t_mfma / (nm16 * 16);
    const double passed = (double)h[4].n_other * per / (nm16 * 16);
    printf("%-26s lds=%d | MFMA alone %6.2f cyc/MFMA | T alone %6.2f cyc/instr | together: %6.2f cyc/MFMA, %5.2f T-instr per MFMA (%.2f cyc of MFMA-wave time per T)\n",
           KIND[T], LDSREAD, m_alone, o_alone, m_both, passed, passed > 0 ? m_both / passed : 0.0);
}

int main(int argc, char** argv)
{
    const int grid = argc > 1 ? atoi(argv[1]) : 256;
    Out* d_out; char* gm;
    hipMalloc(&d_out, sizeof(Out) * grid * 8);
    hipMalloc(&gm, (size_t)grid * 512 * 16 + 4096);
    printf("grid %d workgroups of 8 waves (ticks: s_memtime)\n", grid);
#define R(T) run<T, 0>(d_out, gm, grid); run<T, 1>(d_out, gm, grid);
    R(T_NONE) R(T_MOV) R(T_ADD32) R(T_FMA32) R(T_PKADD32) R(T_PKFMA32) R(T_PKMUL16) R(T_PKMAX16) R(T_CVT) R(T_DPP) R(T_DSW64) R(T_DSR128) R(T_GST64) R(T_MIX) R(T_PKADD16) R(T_CVTRNE) R(T_SWAP) R(T_DSW128) R(T_DSW2) R(T_GST128)
    return 0;
}
