// uva_rdb.hip.h -- the convolutions of 4x_Valar_v1 (`-m r`, models/4x_Valar_v1.param:3-1208,
// upscale/upscale_processing.py:913-916) with the weights STATIONARY in registers.
//
// The layer-by-layer kernel g_conv3_lds (uva_generic.hip.h) feeds its MFMAs from L2 (weights) and LDS (pixels) at
// once and lets several workgroups per CU overlap their phases: 23-34 % matrix-pipe utilisation.  g_conv3_sw keeps a
// wave's share of the weights in its registers for the whole launch and streams the image past them:
//
//   * one persistent 4-wave workgroup per CU, one wave per SIMD.  MBW = 1 (192 -> 64, a dense block's last convolution,
//     with the block's `x*1.0 + conv*0.2` in the epilogue): wave m owns output channels 16m..16m+15 -- 54 k-steps x 4
//     registers = 216 of its 512; MBW = 2 (64 -> 64): wave (h, c) owns channels 32h..32h+31 of the columns 32c..32c+31;
//   * the workgroup walks down a strip of 32 * NWC columns in blocks of FOUR rows.  The six input rows of a block sit
//     in an LDS ring (10 row slots: the next block's four rows are on their way by LDS-DMA meanwhile); a B fragment (16
//     pixels x 32 channels of one input row, one tap column) is ONE conflict-free ds_read_b128 and feeds the MFMAs of
//     up to three output rows (tap rows dy = 0..2) x MBW channel blocks: 0.5 (MBW = 1) / 0.25 reads per MFMA;
//   * 8 (16) independent accumulators per wave, v_mfma_f32_16x16x32_f16, fp32 accumulate; bias, LeakyReLU, the fused
//     element-wise sum and the fp16 conversion exactly as g_conv3_lds does them.
// HBM traffic: the input once (+ 2 columns per strip), the output once.
//
// LDS ring row: [channel chunk of 32][ring column][four 16-byte units], 64 bytes per chunk pixel; unit u of ring column
// rc sits in slot u ^ (((rc >> 2) & 1) << 1): with that XOR the 16 lanes the hardware serves together in a
// ds_read_b128 (lane = (unit << 4) | pixel; groups {0-3,12-15,20-27}, ...) touch 16 different 16-byte bank groups,
// whatever column the fragment starts at (checked by brute force over all origins, tools/lds_swizzle_search.py).
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>

#include "uva_kernels.hip.h"

#ifndef UVA_SW_D
#define UVA_SW_D 3
#endif
#ifndef UVA_SW_DBG
#define UVA_SW_DBG 0
#endif

namespace uva {

struct GSwSeg { int c0, y0, y1, pad; };       // output columns [c0, c0 + SW_C) x rows [y0, y1) of the plane, y1 - y0 a multiple of 4
                                              // except at the plane's bottom

struct GSwArgs {
    const _Float16* in;           // zero-bordered array [(h+3)][(w+2)][in_stride]; pixel (y, x) at row y+1, column x+1
    int in_stride;                // elements per pixel; channels read: [0, 32*KC)
    const half8* wpk;             // pack_generic image (natural octet order): [tap][KC][4][64 lanes][8]
    const float* bias;            // [64]
    _Float16* out;
    int out_stride, out_coff;
    int h, w;
    float slope;                  // LeakyReLU (template ACT)
    // the element-wise sum behind the convolution (GConvArgs::res in uva_generic.hip.h: same expression, same rounding)
    const _Float16* res;
    int res_stride, res_first;
    float ca, cb;
    const GSwSeg* segs;           // this launch's segments; workgroup g owns segs[seg_begin[g] .. seg_begin[g+1])
    const int* seg_begin;
    _Float16* sink;               // >= 64 * 8 bytes: where lanes outside the plane store to
};

constexpr int SW_R = 4;                       // output rows per block
constexpr int SW_SLOTS = 10;                  // ring rows: 6 of the current block + the 4 new ones of the next
template <int MBW> constexpr int sw_cols() { return 32 * (MBW == 1 ? 1 : 2); }       // output columns per strip
template <int KC, int MBW> constexpr int sw_np() { return (KC * (sw_cols<MBW>() + 2) * 4 + 63) / 64; }   // 1-KiB LDS-DMA pieces per ring row
template <int KC, int MBW> constexpr int sw_rowb() { return sw_np<KC, MBW>() * 1024; }
template <int KC, int MBW> constexpr int sw_lds_bytes() { return SW_SLOTS * sw_rowb<KC, MBW>() + 256; }
static_assert(sw_lds_bytes<6, 1>() <= 160 * 1024 && sw_lds_bytes<2, 2>() <= 160 * 1024, "g_conv3_sw LDS budget");

__device__ __forceinline__ void sw_barrier() { asm volatile("s_waitcnt vmcnt(0) lgkmcnt(0)\n\ts_barrier" ::: "memory"); }

// out = x*ca + y*cb of an element-wise sum (BinaryOp ADD, Eltwise SUM with coefficients), rounded to fp16: ONE spelling
// for every kernel that computes it (g_axpby, g_axpby_strided and the convolution epilogues that absorb a sum), so that a
// sum gives the same bytes whichever kernel does it -- left to the compiler, `x*ca + y*cb` contracts into an fma around
// either product
__device__ __forceinline__ _Float16 g_axpby1(float x, float ca, float y, float cb) { return (_Float16)__builtin_fmaf(x, ca, y * cb); }

template <int KC, int MBW, bool ACT>
__global__ __launch_bounds__(256, 1) void g_conv3_sw(GSwArgs a)
{
    constexpr int C = sw_cols<MBW>(), RC = C + 2, NP = sw_np<KC, MBW>(), ROWB = sw_rowb<KC, MBW>();
    constexpr int NF = 2;                      // 16-pixel fragments per wave and row
    constexpr int NIR = SW_R + 2;              // input rows of a block
    constexpr int NPW = (NP + 3) / 4;          // DMA pieces per wave and ring row
    constexpr int NST = SW_R * NF * MBW;       // stores per wave and block: always all of them (lanes outside -> the sink)
    extern __shared__ __attribute__((aligned(16))) char smem[];
    char* const ring = smem;
    float* const lbias = (float*)(smem + SW_SLOTS * ROWB);
    const int wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6), lane = threadIdx.x & 63;
    const int mb0 = MBW == 1 ? wave : 2 * (wave & 1);                // first 16-channel block of this wave
    const int cw = MBW == 1 ? 0 : (wave >> 1) * 32;                  // first strip column of this wave
    const int o = lane >> 4, p = lane & 15;

    // DMA pieces j = wave, wave + 4, ... of a ring row: unit idx = j*64 + lane -> (chunk, ring column, slot) -> source
    // bytes from the row's first pixel (array column c0); the slot holds unit slot ^ swz(rc).  Units past the row's end
    // (padding of the last piece) re-read the last one.
    unsigned voff[NPW];
#pragma unroll
    for (int k = 0; k < NPW; ++k) {
        const int idx = min((wave + 4 * k) * 64 + lane, KC * RC * 4 - 1);
        const int ch = idx / (RC * 4), rem = idx - ch * (RC * 4), rc = rem >> 2, sl = rem & 3;
        const int u = sl ^ (((rc >> 2) & 1) << 1);
        voff[k] = (unsigned)(rc * a.in_stride * 2 + ch * 64 + u * 16);
    }
    if (threadIdx.x < 64) lbias[threadIdx.x] = a.bias ? a.bias[threadIdx.x] : 0.f;

    // per-lane LDS read offsets of the three tap columns (ring column = strip column + dx; cw is a multiple of 32: the
    // swizzle only sees (p + dx) >> 2)
    unsigned offdx[3];
#pragma unroll
    for (int dx = 0; dx < 3; ++dx) {
        const int rc = p + dx;
        offdx[dx] = (unsigned)((cw + rc) * 64 + ((o ^ (((rc >> 2) & 1) << 1)) * 16));
    }
    const unsigned ring_lds = lds_offset(ring);
    const size_t in_pitch = (size_t)(a.w + 2) * a.in_stride;       // elements per array row
    _Float16* const sink = a.sink + lane * 4;

    // ring row rr of a segment holds plane row y0 - 1 + rr = array row y0 + rr (rows below the bottom border: the border
    // row again -- zeros that only feed rows nobody stores) in slot rr % 10
    auto dma_row = [&](int c0, int y0, int rr) {
        const int ay = min(y0 + rr, a.h + 1);
        const char* const src = (const char*)(a.in + (size_t)ay * in_pitch + (size_t)c0 * a.in_stride);
        const unsigned dst = ring_lds + (unsigned)(rr % SW_SLOTS) * ROWB + wave * 1024;
#pragma unroll
        for (int k = 0; k < NPW; ++k)
            if (wave + 4 * k < NP) glds16_s(src, voff[k], dst + k * 4096);
    };

    const int sb = a.seg_begin[blockIdx.x], se = a.seg_begin[blockIdx.x + 1];
    if (sb < se) {        // the first segment's rows are on their way while the weights arrive
        const GSwSeg seg = a.segs[sb];
        for (int rr = 0; rr < NIR; ++rr) dma_row(seg.c0, seg.y0, rr);
    }
    // this wave's weights: k-step (tap, chunk) x its channel blocks
    half8 wgt[9][KC][MBW];
#pragma unroll
    for (int t = 0; t < 9; ++t)
#pragma unroll
        for (int c = 0; c < KC; ++c)
#pragma unroll
            for (int m = 0; m < MBW; ++m) wgt[t][c][m] = a.wpk[((size_t)(t * KC + c) * 4 + mb0 + m) * 64 + lane];

    for (int si = sb; si < se; ++si) {
        const GSwSeg seg = a.segs[si];
        const int c0 = __builtin_amdgcn_readfirstlane(seg.c0), y0 = __builtin_amdgcn_readfirstlane(seg.y0),
                  y1 = __builtin_amdgcn_readfirstlane(seg.y1);
        if (si > sb) {
            sw_barrier();                      // the previous segment's last reads are done
            for (int rr = 0; rr < NIR; ++rr) dma_row(c0, y0, rr);
        }
        sw_barrier();
        const int nblk = (y1 - y0 + SW_R - 1) / SW_R;
        for (int b = 0; b < nblk; ++b) {
            // (1) the next block's four new rows: LDS-DMA, in flight during this block's k-loop
            if (UVA_SW_DBG < 2 && b + 1 < nblk)
                for (int rr = 0; rr < SW_R; ++rr) dma_row(c0, y0, SW_R * (b + 1) + 2 + rr);
            // (2) the fused sum's other operand for this block's pixels: 8 bytes per lane, output row, fragment and block
            size_t pos[SW_R][NF];
            bool inside[SW_R][NF];
            half4 rsv[SW_R][NF][MBW];
#pragma unroll
            for (int r = 0; r < SW_R; ++r)
#pragma unroll
                for (int f = 0; f < NF; ++f) {
                    const int y = y0 + SW_R * b + r, x = c0 + cw + 16 * f + p;
                    inside[r][f] = y < y1 && x < a.w;
                    pos[r][f] = (size_t)(y + 1) * (a.w + 2) + 1 + x;
#pragma unroll
                    for (int m = 0; m < MBW; ++m) {
                        rsv[r][f][m] = half4{(_Float16)0.f, (_Float16)0.f, (_Float16)0.f, (_Float16)0.f};
                        if (UVA_SW_DBG < 1 && a.res && inside[r][f]) rsv[r][f][m] = *(const half4*)(a.res + pos[r][f] * a.res_stride + 16 * (mb0 + m) + 4 * o);
                    }
                }
            // input row ir of the block = ring row 4b + ir
            unsigned rowb[NIR];
#pragma unroll
            for (int ir = 0; ir < NIR; ++ir) rowb[ir] = (unsigned)((SW_R * b + ir) % SW_SLOTS) * ROWB;
            f32x4 acc[SW_R][NF][MBW];         // start from the bias: lane (o, p) accumulates channels 16(mb0 + m) + 4o .. +3
#pragma unroll
            for (int m = 0; m < MBW; ++m) {
                const f32x4 bs = *(const f32x4*)(lbias + 16 * (mb0 + m) + 4 * o);
#pragma unroll
                for (int r = 0; r < SW_R; ++r)
#pragma unroll
                    for (int f = 0; f < NF; ++f) acc[r][f][m] = bs;
            }
            // (3) k-loop: (chunk, tap column) x input row x fragment; fragments are read D steps ahead of their MFMAs
            constexpr int D = UVA_SW_D;
            half8 bq[D + 1][NF];
            auto rd = [&](int idx, half8 (&dst)[NF]) {      // idx = (c * 3 + dx) * NIR + ir
                const int cd = idx / NIR, ir = idx - cd * NIR, c = cd / 3, dx = cd - 3 * c;
#pragma unroll
                for (int f = 0; f < NF; ++f) dst[f] = *(const half8*)(ring + rowb[ir] + offdx[dx] + c * (RC * 64) + f * 1024);
            };
            constexpr int NSTEP = KC * 3 * NIR;
#pragma unroll
            for (int i = 0; i < D; ++i) rd(i, bq[i]);
            __builtin_amdgcn_sched_group_barrier(0x100, D * NF, 0);  // (the pipeline below counts its own reads only)
            static_for<NSTEP>([&](auto I) {
                constexpr int idx = decltype(I)::value;
                constexpr int cd = idx / NIR, ir = idx - cd * NIR, c = cd / 3, dx = cd - 3 * c;
                if constexpr (idx + D < NSTEP) rd(idx + D, bq[(idx + D) % (D + 1)]);
                constexpr int ndy = (ir < 3 ? ir + 1 : 3) - (ir >= SW_R ? ir - SW_R + 1 : 0);     // tap rows with an output row in the block
#pragma unroll
                for (int dy = 0; dy < 3; ++dy) {
                    const int r = ir - dy;
                    if (r < 0 || r >= SW_R) continue;
#pragma unroll
                    for (int f = 0; f < NF; ++f)
#pragma unroll
                        for (int m = 0; m < MBW; ++m)
                            acc[r][f][m] = __builtin_amdgcn_mfma_f32_16x16x32_f16(wgt[dy * 3 + dx][c][m], bq[idx % (D + 1)][f], acc[r][f][m], 0, 0, 0);
                }
                if constexpr (idx + D < NSTEP) __builtin_amdgcn_sched_group_barrier(0x100, NF, 0);
                __builtin_amdgcn_sched_group_barrier(0x008, ndy * NF * MBW, 0);
            });
            // (4) LeakyReLU (ncnn activation_type 2), fp16, the fused sum, 8-byte stores: lane (o, p) holds channels
            // 16(mb0 + m) + 4o .. +3 of pixel (y0 + 4b + r, c0 + cw + 16f + p)
#pragma unroll
            for (int m = 0; m < MBW; ++m) {
                const int ch = 16 * (mb0 + m) + 4 * o;
#pragma unroll
                for (int r = 0; r < SW_R; ++r)
#pragma unroll
                    for (int f = 0; f < NF; ++f) {
                        f32x4 v = acc[r][f][m];
                        if constexpr (ACT) {
#pragma unroll
                            for (int j = 0; j < 4; ++j) v[j] = v[j] > 0.f ? v[j] : v[j] * a.slope;
                        }
                        half4 cv = {(_Float16)v[0], (_Float16)v[1], (_Float16)v[2], (_Float16)v[3]};
                        if (a.res) {
                            const half4 rv = rsv[r][f][m];
                            const half4 xx = a.res_first ? rv : cv, yy = a.res_first ? cv : rv;
#pragma unroll
                            for (int j = 0; j < 4; ++j) cv[j] = g_axpby1((float)xx[j], a.ca, (float)yy[j], a.cb);
                        }
                        _Float16* const dst = (UVA_SW_DBG < 1 && inside[r][f]) ? a.out + pos[r][f] * a.out_stride + a.out_coff + ch : sink;
                        if (UVA_SW_DBG >= 1) { asm volatile("" :: "v"(acc[r][f][m])); continue; }
                        asm volatile("global_store_dwordx2 %0, %1, off" ::"v"(dst), "v"(cv) : "memory");
                    }
            }
            // (5) everything but this block's NST stores has landed (memory operations complete in issue order: the DMA
            // pieces and the loads are older), LDS drained; the stores stay in flight across the barrier
            asm volatile("s_waitcnt vmcnt(%0) lgkmcnt(0)\n\ts_barrier" ::"n"(UVA_SW_DBG >= 1 ? 0 : NST) : "memory");
        }
    }
}

}  // namespace uva
