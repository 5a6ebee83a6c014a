// valu_issue_bench.hip -- what does an instruction cost its wave, alone on its SIMD and beside a wave that streams
// v_mfma_f32_16x16x32_f16?  One workgroup of 8 waves per CU (two per SIMD): waves 0-3 run the probed instruction N times
// (independent chains or one dependent chain), waves 4-7 either idle at a barrier or stream MFMAs meanwhile.
//   hipcc --offload-arch=gfx950 -O3 tools/valu_issue_bench.hip -o /tmp/vib && /tmp/vib
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstring>
typedef _Float16 half8 __attribute__((ext_vector_type(8)));
typedef float f32x4 __attribute__((ext_vector_type(4)));
typedef float f32x2 __attribute__((ext_vector_type(2)));
#define REP 64
#define OUTER 16
template <int OP, int DEP>
__device__ __forceinline__ void probe(float* out, unsigned long long* trace = nullptr)
{
    f32x2 a[8], c = {1.0001f, 0.9999f};
    unsigned u[8];
    for (int i = 0; i < 8; ++i) { a[i] = f32x2{(float)threadIdx.x + i, 1.f}; u[i] = threadIdx.x * 2654435761u + i; }
    extern __shared__ char smem[];
    for (int o = 0; o < OUTER; ++o) {
        if (trace) trace[o] = __builtin_amdgcn_s_memtime();
#pragma unroll
        for (int r = 0; r < REP; ++r) {
            const int k = DEP ? 0 : (r & 7);
            if (OP == 0) asm volatile("v_add_f32 %0, %0, %1" : "+v"(a[k][0]) : "v"(c[0]));
            if (OP == 1) asm volatile("v_pk_add_f32 %0, %0, %1" : "+v"(a[k]) : "v"(c));
            if (OP == 2) asm volatile("v_pk_fma_f32 %0, %0, %1, %1" : "+v"(a[k]) : "v"(c));
            if (OP == 3) asm volatile("v_med3_f32 %0, %0, %1, %2" : "+v"(a[k][0]) : "v"(c[0]), "v"(c[1]));
            if (OP == 4) asm volatile("v_cvt_pk_f16_f32 %0, %1, %2" : "=v"(u[k]) : "v"(a[k][0]), "v"(a[k][1]));
            if (OP == 5) asm volatile("v_mov_b32_dpp %0, %1 row_shl:1 row_mask:0xf bank_mask:0xf bound_ctrl:1" : "=v"(u[k]) : "v"(u[(k + 1) & 7]));
            if (OP == 6) asm volatile("v_pk_add_f16 %0, %0, %1" : "+v"(u[k]) : "v"(u[(k + 1) & 7]));
            if (OP == 7) asm volatile("v_permlane16_swap_b32 %0, %1" : "+v"(u[k]), "+v"(u[(k + 4) & 7]));
            if (OP == 8) asm volatile("ds_write_b128 %0, %1" :: "v"((unsigned)(threadIdx.x & 63) * 16 + (k << 10)), "v"(*(f32x4*)&a[k & 6]) : "memory");
            if (OP == 9) asm volatile("ds_read_b128 %0, %1" : "=v"(*(f32x4*)&a[k & 6]) : "v"((unsigned)(threadIdx.x & 63) * 16 + (k << 10)) : "memory");
            if (OP == 10) asm volatile("v_pk_mul_f32 %0, %0, %1" : "+v"(a[k]) : "v"(c));
            if (OP == 11) asm volatile("ds_write_b64 %0, %1" :: "v"((unsigned)(threadIdx.x & 63) * 8 + (k << 10)), "v"(a[k]) : "memory");
            if (OP == 12) asm volatile("ds_write_b32 %0, %1" :: "v"((unsigned)(threadIdx.x & 63) * 4 + (k << 10)), "v"(u[k]) : "memory");
            if (OP == 13) asm volatile("ds_write2_b64 %0, %1, %2 offset1:64" :: "v"((unsigned)(threadIdx.x & 63) * 8 + (k << 10)), "v"(a[k]), "v"(a[(k + 1) & 7]) : "memory");
            if (OP == 15) { asm volatile("v_add_f32 %0, %0, %1" : "+v"(a[k][0]) : "v"(c[0])); if ((r & 3) == 3) asm volatile("s_nop 7\n s_nop 7"); }
            if (OP == 14) asm volatile("ds_write_b128 %0, %1" :: "v"((unsigned)(threadIdx.x & 63) * 16 + (threadIdx.x >> 6) * 8192 + (k << 10)), "v"(*(f32x4*)&a[k & 6]) : "memory");
        }
    }
    asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
    float s = 0;
    for (int i = 0; i < 8; ++i) s += a[i][0] + a[i][1] + (float)u[i];
    if (s == 123.456f) out[0] = s;
}
template <int OP, int DEP, int MFMA, int PRIO = 0, int NW = 8>
__global__ __launch_bounds__(64 * NW, 1) void k(unsigned long long* t, float* out)
{
    const int wave = threadIdx.x >> 6;
    __syncthreads();
    const unsigned long long t0 = __builtin_amdgcn_s_memtime();
    if (wave < 4 || wave >= 8) {
        if (PRIO && PRIO < 4) __builtin_amdgcn_s_setprio(PRIO);
        probe<OP, DEP>(out, (PRIO == 9 && threadIdx.x == 0 && blockIdx.x == 0) ? t + 16 : nullptr);
    } else if (MFMA) {
        half8 x = {1, 2, 3, 4, 5, 6, 7, 8}, y = {8, 7, 6, 5, 4, 3, 2, 1};
        f32x4 acc[8] = {};
        for (int o = 0; o < OUTER * MFMA; ++o)
#pragma unroll
            for (int r = 0; r < REP; ++r) acc[r & 7] = __builtin_amdgcn_mfma_f32_16x16x32_f16(x, y, acc[r & 7], 0, 0, 0);
        float s = 0;
        for (int i = 0; i < 8; ++i) s += acc[i][0];
        if (s == 123.456f) out[1] = s;
    }
    const unsigned long long t1 = __builtin_amdgcn_s_memtime();
    if ((threadIdx.x & 63) == 0 && blockIdx.x == 0) t[wave] = t1 - t0;
    if (PRIO == 9 && threadIdx.x == 0 && blockIdx.x == 0) { t[40] = t0; t[41] = t1; }
    if (PRIO == 9 && threadIdx.x == 256 && blockIdx.x == 0) { t[42] = t0; t[43] = t1; }
}
template <int OP, int DEP, int MFMA, int PRIO = 0, int NW = 8>
void run(const char* name, unsigned long long* d, float* o)
{
    unsigned long long h[12];
    k<OP, DEP, MFMA, PRIO, NW><<<256, 64 * NW, 65536>>>(d, o);
    k<OP, DEP, MFMA, PRIO, NW><<<256, 64 * NW, 65536>>>(d, o);
    hipDeviceSynchronize();
    hipMemcpy(h, d, sizeof h, hipMemcpyDeviceToHost);
    const double n = (double)REP * OUTER;
    printf("%-22s %s %s : probe wave %6.1f cycles/instr   mfma wave %6.1f cycles/mfma\n", name, DEP ? "dependent  " : "independent", MFMA ? "beside MFMA" : "alone      ",
           h[0] / n, MFMA ? h[4] / (n * MFMA) : 0.0);
}
int main()
{
    unsigned long long* d; float* o;
    hipMalloc(&d, 512); hipMalloc(&o, 64);
    hipFuncSetAttribute((const void*)k<0, 0, 0>, hipFuncAttributeMaxDynamicSharedMemorySize, 65536);
#define ALL(OP, NAME) run<OP, 0, 0>(NAME, d, o); run<OP, 1, 0>(NAME, d, o); run<OP, 0, 1>(NAME, d, o); run<OP, 1, 1>(NAME, d, o);
    ALL(0, "v_add_f32") ALL(1, "v_pk_add_f32") ALL(2, "v_pk_fma_f32") ALL(10, "v_pk_mul_f32") ALL(3, "v_med3_f32") ALL(4, "v_cvt_pk_f16_f32")
    ALL(5, "v_mov_b32_dpp") ALL(6, "v_pk_add_f16") ALL(7, "v_permlane16_swap") ALL(8, "ds_write_b128") ALL(9, "ds_read_b128")
    run<0, 0, 4>("v_add_f32 (mfma x4)", d, o);
    ALL(11, "ds_write_b64") ALL(12, "ds_write_b32") ALL(13, "ds_write2_b64") ALL(14, "ds_write_b128 per-wave 8K")
    // the probe waves at a raised priority; and TWO probe waves beside every MFMA wave (12 waves per CU)
    run<0, 0, 1, 3>("v_add_f32 prio 3", d, o); run<0, 1, 1, 3>("v_add_f32 prio 3", d, o); run<6, 0, 1, 3>("v_pk_add_f16 prio 3", d, o);
    run<0, 0, 2, 0, 12>("v_add_f32 2 probes", d, o); run<0, 1, 2, 0, 12>("v_add_f32 2 probes", d, o); run<0, 0, 0, 0, 12>("v_add_f32 2 probes", d, o);
    // progress of a probe wave beside an MFMA stream four times as long: when do its instructions get through?
    run<0, 0, 4, 9>("v_add_f32 traced", d, o);
    { unsigned long long tr[17]; hipMemcpy(tr, d + 16, sizeof tr, hipMemcpyDeviceToHost);
      unsigned long long e[4]; hipMemcpy(e, d + 40, sizeof e, hipMemcpyDeviceToHost);
      printf("  cycles per 64 v_add_f32, sixteen times in a row:"); for (int i = 1; i < 16; ++i) printf(" %llu", tr[i] - tr[i - 1]);
      printf("\n  probe wave: t0 -> first stamp %llu, last stamp -> t1 %llu; MFMA wave 4: starts %lld after the probe wave, ends %lld after the probe wave's t1\n",
             tr[0] - e[0], e[1] - tr[15], (long long)(e[2] - e[0]), (long long)(e[3] - e[1])); }
    run<15, 0, 1>("v_add_f32, 1 in 4 + nops", d, o);
    run<9, 0, 1, 3>("ds_read_b128 prio 3", d, o); run<8, 0, 1, 3>("ds_write_b128 prio 3", d, o);
    return 0;
}
