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
#include "uva_sw.h"

#define UVA_SW_D 3
#define UVA_RA_D 2       // rdb4_kernel: k-steps (of 3 fragment reads) the reads run ahead of the MFMAs
#define UVA_SK_DBG 0      // g_conv3_sk timing experiments only (results are wrong): 1 no epilogue, 2 no exchange, 4 no DMA
#define UVA_SW_HALFREAD 0
#define UVA_SW_DBG 0
#define UVA_SW_DBG_X 0     // CEILING EXPERIMENT (wrong results), with UVA_RA_DBG=1: g_conv3_sw<6,..> fetches the channels behind the first 64
                              // (x1..x4 of a dense block) from ONE fixed array row, i.e. from L2: together "x1..x4 never cross HBM"
#define UVA_RA_DBG 0       // timing experiments only (results are wrong): 1 no HBM stores, 2 no epilogue, 4 no x DMA, 8 no hand-over

namespace uva {

constexpr int SW_SLOTS = 10;                  // ring rows: 6 of the current block + the 4 new ones of the next
template <int MBW> constexpr int sw_cols() { return 32 * (MBW == 1 ? 1 : 2); }       // output columns per strip
template <int KC, int MBW> constexpr int sw_np() { return (KC * (sw_cols<MBW>() + 2) * 4 + 63) / 64; }   // 1-KiB LDS-DMA pieces per ring row
template <int KC, int MBW> constexpr int sw_rowb() { return sw_np<KC, MBW>() * 1024; }
template <int KC, int MBW> constexpr int sw_lds_bytes() { return SW_SLOTS * sw_rowb<KC, MBW>() + 256; }
static_assert(sw_lds_bytes<6, 1>() <= 160 * 1024 && sw_lds_bytes<2, 2>() <= 160 * 1024, "g_conv3_sw LDS budget");

__device__ __forceinline__ void sw_barrier() { asm volatile("s_waitcnt vmcnt(0) lgkmcnt(0)\n\ts_barrier" ::: "memory"); }

// RES / RES2: the first / second fused sum -- 0 none, 1 the other operand is the sum's x (out = res*ca + conv*cb), 2 the
// convolution's side is (out = conv*ca + res*cb); g_axpby1 is not symmetric in x and y, so the order is part of the bytes
// UP: the input is the nearest-neighbour 2x enlargement (ncnn Interp, resize_type 1) of the array `in` points at: ring
// column / row r of the enlarged plane's array is column / row (r + 1) >> 1 of the source's (borders included: 0 -> 0,
// 2n + 1 -> n + 1), so the row DMA just fetches from there -- every lane of an LDS-DMA has its own address -- and the
// enlarged array (16x the bytes of the 1x plane at 4x) is never written or read.  ph, pw stay the OUTPUT plane's size.
template <int KC, int MBW, bool ACT, int RES, int RES2, bool UP = false>
__global__ __launch_bounds__(256, 1) void g_conv3_sw(GSwArgs a)
{
    constexpr int C = sw_cols<MBW>(), RC = C + 2, NP = sw_np<KC, MBW>(), ROWB = sw_rowb<KC, MBW>();
    constexpr int NF = 2;                      // 16-pixel fragments per wave and row
    constexpr int NIR = SW_R + 2;              // input rows of a block
    constexpr int NPW = (NP + 3) / 4;          // DMA pieces per wave and ring row
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
        voff[k] = (unsigned)((UP ? (rc + 1) >> 1 : rc) * a.in_stride * 2 + ch * 64 + u * 16);        // (UP: c0 is even)
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
    _Float16* const sink = a.sink + lane * 4;
    // the plane of the current segment
    const _Float16* pin = nullptr;
    _Float16* pout = nullptr;
    const _Float16* pres = nullptr;
    const _Float16* pres2 = nullptr;
    int ph = 0, pw = 0;
    size_t in_pitch = 0;                                           // elements per array row
    auto set_plane = [&](int pl) {
        pin = a.in[pl]; pout = a.out[pl]; pres = a.res[pl]; pres2 = a.res2[pl];
        ph = a.ph[pl]; pw = a.pw[pl];
        in_pitch = (size_t)((UP ? pw >> 1 : pw) + 2) * a.in_stride;
    };

    // ring row rr of a segment holds plane row y0 - 1 + rr = array row y0 + rr (rows below the bottom border: the border
    // row again -- zeros that only feed rows nobody stores) in slot rr % 10
    auto dma_row = [&](int c0, int y0, int rr) {
        const int ayo = min(y0 + rr, ph + 1), ay = UP ? (ayo + 1) >> 1 : ayo;
        const char* const src = (const char*)(pin + (size_t)ay * in_pitch + (size_t)(UP ? c0 >> 1 : c0) * a.in_stride);
        const unsigned dst = ring_lds + (unsigned)(rr % SW_SLOTS) * ROWB + wave * 1024;
        if constexpr (UVA_SW_DBG_X != 0 && KC == 6) {
            const char* const src2 = (const char*)(pin + (size_t)1 * in_pitch + (size_t)c0 * a.in_stride);      // row 1 whatever the block
#pragma unroll
            for (int k = 0; k < NPW; ++k)
                if (wave + 4 * k < NP) {
                    const int idx = min((wave + 4 * k) * 64 + lane, KC * RC * 4 - 1);
                    glds16((idx / (RC * 4) >= 2 ? src2 : src) + voff[k], dst + k * 4096);
                }
            return;
        }
#pragma unroll
        for (int k = 0; k < NPW; ++k)
            if (wave + 4 * k < NP) glds16_s(src, voff[k], dst + k * 4096);
    };

    const int sb = a.seg_begin[blockIdx.x], se = a.seg_begin[blockIdx.x + 1];
    if (sb < se) {        // the first segment's rows are on their way while the weights arrive
        const GSwSeg seg = a.segs[sb];
        set_plane(__builtin_amdgcn_readfirstlane(seg.plane));
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

    // The epilogue of block b (LeakyReLU, fp16, the fused sums, the stores) runs INSIDE the k-loop of block b+1, one tile
    // every few steps: a wave has its SIMD to itself, so whatever is not between MFMAs is matrix-pipe idle time.  Two
    // accumulator sets alternate; the sums' other operands are fetched tile by tile, each right after the previous
    // block's value in the same registers has been used (a whole k-loop ahead of its own use).
    constexpr int NT = SW_R * NF * MBW;        // tiles per wave and block
    constexpr int NLD = NT * ((RES != 0) + (RES2 != 0));
    constexpr int NSTEP = KC * 3 * NIR;
    constexpr int E0 = NSTEP * 2 / 5, ES = (NSTEP - 2 - E0) / NT > 0 ? (NSTEP - 2 - E0) / NT : 1;   // tile j after step E0 + j*ES
    static_assert(E0 + (NT - 1) * ES < NSTEP, "epilogue tiles fit the k-loop");
    half4 rsv[RES ? NT : 1], rsw[RES2 ? NT : 1];

    // tile j = (r, f, m) of block b of the segment (c0, y0, y1): is this lane's pixel a real one, where it sits in the arrays
    auto tile_pos = [&](int j, int c0, int y0, int y1, int b, bool* inside) -> size_t {
        const int f = (j / MBW) % NF, r = j / (MBW * NF);
        const int y = y0 + SW_R * b + r, x = c0 + cw + 16 * f + p;
        *inside = y < y1 && x < pw;
        return ((size_t)(min(y, ph - 1) + 1) * (pw + 2) + 1 + min(x, pw - 1));      // (clamped: lanes outside read a real pixel)
    };
    auto load_res = [&](int j, int c0, int y0, int y1, int b) {
        if constexpr (RES != 0 || RES2 != 0) {
            bool in;
            const size_t pos = tile_pos(j, c0, y0, y1, b, &in);
            const int ch = 16 * (mb0 + j % MBW) + 4 * o;
            if constexpr (RES != 0) rsv[j] = *(const half4*)(pres + pos * a.res_stride + ch);
            if constexpr (RES2 != 0) rsw[j] = *(const half4*)(pres2 + pos * a.res2_stride + ch);
        }
    };
    auto epi_tile = [&](auto Jc, const f32x4 (&acc)[SW_R][NF][MBW], int c0, int y0, int y1, int b) {
        constexpr int j = decltype(Jc)::value, m = j % MBW, f = (j / MBW) % NF, r = j / (MBW * NF);
        f32x4 v = acc[r][f][m];
        if constexpr (ACT) {
#pragma unroll
            for (int e = 0; e < 4; ++e) v[e] = v[e] > 0.f ? v[e] : v[e] * a.slope;
        }
        half4 cv = {(_Float16)v[0], (_Float16)v[1], (_Float16)v[2], (_Float16)v[3]};
        if constexpr (RES != 0) {
            const half4 rv = rsv[j];
#pragma unroll
            for (int e = 0; e < 4; ++e)
                cv[e] = RES == 1 ? g_axpby1((float)rv[e], a.ca, (float)cv[e], a.cb) : g_axpby1((float)cv[e], a.ca, (float)rv[e], a.cb);
        }
        if constexpr (RES2 != 0) {
            const half4 rv = rsw[j];
#pragma unroll
            for (int e = 0; e < 4; ++e)
                cv[e] = RES2 == 1 ? g_axpby1((float)rv[e], a.ca2, (float)cv[e], a.cb2) : g_axpby1((float)cv[e], a.ca2, (float)rv[e], a.cb2);
        }
        bool in;
        const size_t pos = tile_pos(j, c0, y0, y1, b, &in);
        _Float16* const dst = (UVA_SW_DBG < 1 && in) ? pout + pos * a.out_stride + a.out_coff + 16 * (mb0 + m) + 4 * o : sink;
        if (UVA_SW_DBG >= 1) { asm volatile("" ::"v"(cv)); return; }
        asm volatile("global_store_dwordx2 %0, %1, off" ::"v"(dst), "v"(cv) : "memory");
    };
    // the k-loop of block b into acc: (chunk, tap column) x input row x fragment, fragments read D steps ahead; side(j)
    // = tile j of the previous block's epilogue
    auto kloop = [&](f32x4 (&acc)[SW_R][NF][MBW], const int b, auto&& side) {
        unsigned rowb[NIR];                    // input row ir of the block = ring row 4b + ir
#pragma unroll
        for (int ir = 0; ir < NIR; ++ir) rowb[ir] = (unsigned)((SW_R * b + ir) % SW_SLOTS) * ROWB;
#pragma unroll
        for (int m = 0; m < MBW; ++m) {        // start from the bias: lane (o, p) accumulates channels 16(mb0 + m) + 4o .. +3
            const f32x4 bs = *(const f32x4*)(lbias + 16 * (mb0 + m) + 4 * o);
#pragma unroll
            for (int r = 0; r < SW_R; ++r)
#pragma unroll
                for (int f = 0; f < NF; ++f) acc[r][f][m] = bs;
        }
        constexpr int D = UVA_SW_D;
        half8 bq[D + 1][NF];
        auto rd = [&](int idx, half8 (&dst)[NF]) {      // idx = (c * 3 + dx) * NIR + ir
            const int cd = idx / NIR, ir = idx - cd * NIR, c = cd / 3, dx = cd - 3 * c;
#pragma unroll
            for (int f = 0; f < NF; ++f) {
                if (UVA_SW_HALFREAD && f > 0) { dst[f] = dst[0]; continue; }       // timing experiment only (results are wrong)
                dst[f] = *(const half8*)(ring + rowb[ir] + offdx[dx] + c * (RC * 64) + f * 1024);
            }
        };
        __builtin_amdgcn_sched_barrier(0);                       // nothing else's LDS reads count as the pipeline's
#pragma unroll
        for (int i = 0; i < D; ++i) rd(i, bq[i]);
        __builtin_amdgcn_sched_group_barrier(0x100, D * (UVA_SW_HALFREAD ? 1 : NF), 0);
        static_for<NSTEP>([&](auto I) {
            constexpr int idx = decltype(I)::value;
            constexpr int cd = idx / NIR, ir = idx - cd * NIR, c = cd / 3, dx = cd - 3 * c;
            if constexpr (idx + D < NSTEP) rd(idx + D, bq[(idx + D) % (D + 1)]);
            constexpr int ndy = (ir < 3 ? ir + 1 : 3) - (ir >= SW_R ? ir - SW_R + 1 : 0);     // tap rows with an output row in the block
            // fragment-major: a B fragment feeds its tap rows and channel blocks back to back (neighbours that share the B operand
            // draw less at the power cap than neighbours that share the weights: tools/mfma_operand_order_bench.hip; every
            // accumulator still sees its own additions in the same order: the same bytes)
#pragma unroll
            for (int f = 0; f < NF; ++f) {
#pragma unroll
                for (int dy = 0; dy < 3; ++dy) {
                    const int r = ir - dy;
                    if (r < 0 || r >= SW_R) continue;
#pragma unroll
                    for (int m = 0; m < MBW; ++m)
                        acc[r][f][m] = __builtin_amdgcn_mfma_f32_16x16x32_f16(wgt[dy * 3 + dx][c][m], bq[idx % (D + 1)][f], acc[r][f][m], 0, 0, 0);
                }
            }
            if constexpr (idx + D < NSTEP) __builtin_amdgcn_sched_group_barrier(0x100, UVA_SW_HALFREAD ? 1 : NF, 0);
            __builtin_amdgcn_sched_group_barrier(0x008, ndy * NF * MBW, 0);
            if constexpr (idx >= E0 && (idx - E0) % ES == 0 && (idx - E0) / ES < NT) side(std::integral_constant<int, (idx - E0) / ES>{});
        });
        __builtin_amdgcn_sched_barrier(0);
    };

    for (int si = sb; si < se; ++si) {
        const GSwSeg seg = a.segs[si];
        const int c0 = __builtin_amdgcn_readfirstlane(seg.c0), y0 = __builtin_amdgcn_readfirstlane(seg.y0),
                  y1 = __builtin_amdgcn_readfirstlane(seg.y1);
        if (si > sb) {
            sw_barrier();                      // the previous segment's last reads are done
            set_plane(__builtin_amdgcn_readfirstlane(seg.plane));
            for (int rr = 0; rr < NIR; ++rr) dma_row(c0, y0, rr);
        }
        sw_barrier();
        const int nblk = (y1 - y0 + SW_R - 1) / SW_R;
        f32x4 accA[SW_R][NF][MBW], accB[SW_R][NF][MBW];
        // one iteration: the DMA of block b+1's rows, the k-loop of block b into `cur`, the epilogue of block b-1 from `prev`
        // (its tiles' other operands replaced by block b's as they are used).  Exactly NT stores and NLD loads follow the
        // DMA pieces, so "all but the newest NT + NLD memory operations have completed" proves the DMA (operations
        // complete in issue order) and leaves the stores and the loads in flight across the barrier.
        auto iteration = [&](f32x4 (&cur)[SW_R][NF][MBW], f32x4 (&prev)[SW_R][NF][MBW], const int b) {
            if (UVA_SW_DBG < 2)
                for (int rr = 0; rr < SW_R; ++rr) dma_row(c0, y0, SW_R * (b + 1) + 2 + rr);      // (past the last block: rows nobody reads)
            kloop(cur, b, [&](auto J) {
                epi_tile(J, prev, c0, y0, y1, b - 1);
                load_res(decltype(J)::value, c0, y0, y1, b);
            });
            asm volatile("s_waitcnt vmcnt(%0) lgkmcnt(0)\n\ts_barrier" ::"n"(UVA_SW_DBG >= 1 ? 0 : NT + NLD) : "memory");
        };
        // block 0: nothing to finish yet; its operands are fetched up front
        if (UVA_SW_DBG < 2)
            for (int rr = 0; rr < SW_R; ++rr) dma_row(c0, y0, SW_R + 2 + rr);
#pragma unroll
        for (int j = 0; j < NT; ++j) load_res(j, c0, y0, y1, 0);
        kloop(accA, 0, [](auto) {});
        asm volatile("s_waitcnt vmcnt(%0) lgkmcnt(0)\n\ts_barrier" ::"n"(UVA_SW_DBG >= 1 ? 0 : NLD) : "memory");
        int b = 1;
        for (; b + 1 < nblk; b += 2) {
            iteration(accB, accA, b);
            iteration(accA, accB, b + 1);
        }
        if (b < nblk) {
            iteration(accB, accA, b);
            static_for<NT>([&](auto J) { epi_tile(J, accB, c0, y0, y1, nblk - 1); });
        } else {
            static_for<NT>([&](auto J) { epi_tile(J, accA, c0, y0, y1, nblk - 1); });
        }
    }
}

// ---------------------------------------------------------------------------------------------------------------
// g_conv3_sk: the 192 -> 64 convolution (a dense block's last one, with its fused sums) on v_mfma_f32_32x32x16_f16,
// the k-loop SPLIT between wave pairs.  UVA_GENERIC_SK=1 selects it instead of g_conv3_sw<6, 1>.
//
// Built to find out what bounds g_conv3_sw<6, 1> (matrix pipes half busy): half the MFMA instructions for the same
// flops, half the fragment reads per flop of the 16x16x32 version of this kernel, the sums' operands waited for with
// exact counts.  Result, same box: 438 - 448 us against g_conv3_sw<6, 1>'s 440 - 449 -- three structures, one time;
// the launch sits at the package's power limit (1 335 - 1 360 W, 2.07 - 2.10 GHz) and what is removed in cycles comes
// back as clock only where it also removes energy (profiles/r03_ab_results.txt, blocks 9 - 11).  The default stays the
// simpler kernel; this one is kept correct (tests/test_generic_graph.py runs it) as the starting point for work that
// removes energy: fewer wasted columns, the block's last convolution fed from LDS instead of HBM.
//
// Wave (mh, kh) owns 32 output channels (one 32-row MFMA block) and HALF the input channels (chunks 3kh .. 3kh + 2: 216
// weight registers): a k-step is (chunk, tap column, input row) = two B fragments of 16 channels x 32 pixels (the whole
// strip width), each feeding the MFMAs of up to three output rows -- 0.5 reads per 32-cycle MFMA.  The reduction: after a
// block's k-loop the pair exchanges two of its four 32 x 32 partial tiles through LDS (fp32); wave kh finishes rows 2kh,
// 2kh + 1 of the block (its own partial + the partner's; kh = 0 started from the bias) and does their epilogue inside
// the next block's k-loop.
//
// LDS: ring rows packed (34 columns x 6 chunks x 64 B = 13 056 B, the 13th DMA piece is 48 lanes wide), unit u of ring
// column rc in slot u ^ ((rc >> 2) & 3) -- lane l reads unit 2h + (l >> 5) of column (l & 31) + dx, conflict-free for
// every origin under the ds_read_b128 lane groups (tools/lds_swizzle_search.py); 10 x 13 056 + 4 x 8 KiB of exchange
// buffers + the bias = 163 584 B of the 163 840.
// ---------------------------------------------------------------------------------------------------------------
typedef float f32x16 __attribute__((ext_vector_type(16)));
constexpr int SK_KC = 6, SK_KH = 3, SK_C = 32, SK_RC = SK_C + 2;
constexpr int SK_UNITS = SK_KC * SK_RC * 4;                 // 16-byte units of a ring row
constexpr int SK_ROWB = SK_UNITS * 16;
constexpr int SK_NP = (SK_UNITS + 63) / 64;                 // DMA pieces per ring row; the last one has SK_UNITS % 64 lanes
constexpr int SK_HAND = SW_SLOTS * SK_ROWB, SK_BIAS = SK_HAND + 4 * 8 * 1024;
constexpr int sk_lds_bytes() { return SK_BIAS + 256; }
static_assert(sk_lds_bytes() <= 160 * 1024 && SK_ROWB % 256 == 0, "g_conv3_sk LDS budget / bank alignment of the rows");

template <int RES, int RES2>
__global__ __launch_bounds__(256, 1) void g_conv3_sk(GSwArgs a)
{
    constexpr int KC = SK_KC, KH = SK_KH, RC = SK_RC, ROWB = SK_ROWB, NP = SK_NP;
    constexpr int NIR = SW_R + 2, NPW = (NP + 3) / 4;
    constexpr int NT = 2 * 4;                  // pieces a wave FINISHES per block: 2 rows x 4 quads of channels (8 bytes per lane)
    constexpr int NLD = NT * ((RES != 0) + (RES2 != 0));
    extern __shared__ __attribute__((aligned(16))) char smem[];
    char* const ring = smem;
    char* const hand = smem + SK_HAND;
    float* const lbias = (float*)(smem + SK_BIAS);
    const int wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6), lane = threadIdx.x & 63;
    const int mh = wave & 1, khw = wave >> 1;
    const int g5 = lane >> 5, p = lane & 31;   // MFMA 32x32x16: B column / D column = pixel p; k octet (B) and row quad (D) by g5

    unsigned voff[NPW];
#pragma unroll
    for (int k = 0; k < NPW; ++k) {
        const int idx = min((wave + 4 * k) * 64 + lane, SK_UNITS - 1);
        const int ch = idx / (RC * 4), rem = idx - ch * (RC * 4), rc = rem >> 2, sl = rem & 3;
        const int u = sl ^ ((rc >> 2) & 3);
        voff[k] = (unsigned)(rc * a.in_stride * 2 + ch * 64 + u * 16);
    }
    if (threadIdx.x < 64) lbias[threadIdx.x] = a.bias ? a.bias[threadIdx.x] : 0.f;
    // per-lane LDS read offsets: tap column dx, channel half h of a chunk (unit 2h + g5 of ring column p + dx)
    unsigned offr[3][2];
#pragma unroll
    for (int dx = 0; dx < 3; ++dx)
#pragma unroll
        for (int h = 0; h < 2; ++h) {
            const int rc = p + dx;
            offr[dx][h] = (unsigned)(rc * 64 + (((2 * h + g5) ^ ((rc >> 2) & 3)) * 16) + khw * KH * (RC * 64));
        }
    const unsigned ring_lds = lds_offset(ring);
    _Float16* const sink = a.sink + lane * 4;
    const _Float16* pin = nullptr;
    _Float16* pout = nullptr;
    const _Float16* pres = nullptr;
    const _Float16* pres2 = nullptr;
    int ph = 0, pw = 0;
    size_t in_pitch = 0;
    auto set_plane = [&](int pl) {
        pin = a.in[pl]; pout = a.out[pl]; pres = a.res[pl]; pres2 = a.res2[pl];
        ph = a.ph[pl]; pw = a.pw[pl];
        in_pitch = (size_t)(pw + 2) * a.in_stride;
    };
    auto dma_row = [&](int c0, int y0, int rr) {
        const int ay = min(y0 + rr, ph + 1);
        const unsigned long long sa = (unsigned long long)(pin + (size_t)ay * in_pitch + (size_t)c0 * a.in_stride);
        // (made uniform by hand: the last piece sits under a lane mask, and an address the compiler has sunk into that
        // divergent region does not reach the instruction's scalar operand)
        const char* const src = (const char*)(((unsigned long long)(unsigned)__builtin_amdgcn_readfirstlane((unsigned)(sa >> 32)) << 32) |
                                              (unsigned)__builtin_amdgcn_readfirstlane((unsigned)sa));       // (the builtin returns int)
        const unsigned dst = __builtin_amdgcn_readfirstlane(ring_lds + (unsigned)(rr % SW_SLOTS) * ROWB + wave * 1024);
#pragma unroll
        for (int k = 0; k < NPW; ++k) {
            if (4 * k + 3 < NP - 1) glds16_s(src, voff[k], dst + k * 4096);           // (a piece every wave has, all 64 lanes)
            else if (wave + 4 * k < NP - 1) glds16_s(src, voff[k], dst + k * 4096);
            else if (wave + 4 * k == NP - 1 && lane < SK_UNITS - (NP - 1) * 64) glds16_s(src, voff[k], dst + k * 4096);
        }
    };

    const int sb = a.seg_begin[blockIdx.x], se = a.seg_begin[blockIdx.x + 1];
    if (sb < se) {
        const GSwSeg seg = a.segs[sb];
        set_plane(__builtin_amdgcn_readfirstlane(seg.plane));
        for (int rr = 0; rr < NIR; ++rr) dma_row(seg.c0, seg.y0, rr);
    }
    // everything below once per k-half: a wave's accumulator rows are rotated so that acc[0], acc[1] are the rows it
    // finishes (2kh, 2kh + 1) -- static register indices on both paths, no selects
    auto run = [&](auto KHc) {
    constexpr int kh = decltype(KHc)::value;
    // A fragments out of pack_generic's 16x16x32 image: row p of the 32-row block is row p & 15 of 16-row block (p >> 4),
    // k octet 2h + g5 of the chunk
    half8 wgt[9][KH][2];
#pragma unroll
    for (int t = 0; t < 9; ++t)
#pragma unroll
        for (int c = 0; c < KH; ++c)
#pragma unroll
            for (int h = 0; h < 2; ++h)
                wgt[t][c][h] = a.wpk[((size_t)(t * KC + kh * KH + c) * 4 + 2 * mh + (p >> 4)) * 64 + ((2 * h + g5) * 16 + (p & 15))];

    constexpr int NSTEP = KH * 3 * NIR;
    constexpr int E0 = NSTEP / 4, ES = (NSTEP - 2 - E0) / NT > 0 ? (NSTEP - 2 - E0) / NT : 1;
    static_assert(E0 + (NT - 1) * ES < NSTEP, "epilogue pieces fit the k-loop");
    half4 rsv[RES ? NT : 1], rsw[RES2 ? NT : 1];

    // finished piece j = (rr, q) of block b: row 4b + 2kh + rr, this lane's pixel, channels 32mh + 8q + 4g5 .. + 3
    // (accumulator registers 4q .. 4q + 3 of the row's 32 x 32 tile)
    auto piece_pos = [&](int j, int c0, int y0, int y1, int b, bool* inside) -> size_t {
        const int rr = j >> 2;
        const int y = y0 + SW_R * b + 2 * kh + rr, x = c0 + p;
        *inside = (y < y1) & (x < pw);
        return ((size_t)(min(y, ph - 1) + 1) * (pw + 2) + 1 + min(x, pw - 1));
    };
    const int chq = 32 * mh + 4 * g5;
    auto load_res = [&](int j, int c0, int y0, int y1, int b) {
        if constexpr (RES != 0 || RES2 != 0) {
            bool in;
            const size_t pos = piece_pos(j, c0, y0, y1, b, &in);
            const int ch = chq + 8 * (j & 3);
            // (asm: the compiler cannot count the LDS-DMA pieces and the stores, both asm, that sit between a load and its
            // use -- its own s_waitcnt vmcnt(7) in front of every use waited for the DMA issued at the top of the block,
            // 1.5 us per block; the waits are epi_piece's)
            if constexpr (RES != 0) {
                const _Float16* const q1 = pres + pos * a.res_stride + ch;
                asm volatile("global_load_dwordx2 %0, %1, off" : "=v"(rsv[j]) : "v"(q1));
            }
            if constexpr (RES2 != 0) {
                const _Float16* const q2 = pres2 + pos * a.res2_stride + ch;
                asm volatile("global_load_dwordx2 %0, %1, off" : "=v"(rsw[j]) : "v"(q2));
            }
        }
    };
    // vector-memory operations a wave issues between the load of piece j's operands and their use one block later, if
    // nothing else intervenes: the rest of that k-loop's pieces (one store + NRL loads each), the next block's DMA pieces
    // (12; wave 0 has 16 and waits for four of them), the pieces before j of the current k-loop
    constexpr int NRL = (RES != 0) + (RES2 != 0);
    constexpr int KW_LOOP = (NT - 1) * (1 + NRL) + 4 * (NPW - 1);
    auto epi_piece = [&](auto Jc, auto KWc, const f32x16 (&fin)[2], int c0, int y0, int y1, int b) {
        constexpr int j = decltype(Jc)::value, q = j & 3, rr = j >> 2;
        if constexpr (RES != 0 && RES2 != 0) asm volatile("s_waitcnt vmcnt(%2)" : "+v"(rsv[j]), "+v"(rsw[j]) : "n"(decltype(KWc)::value));
        else if constexpr (RES != 0) asm volatile("s_waitcnt vmcnt(%1)" : "+v"(rsv[j]) : "n"(decltype(KWc)::value));
        else if constexpr (RES2 != 0) asm volatile("s_waitcnt vmcnt(%1)" : "+v"(rsw[j]) : "n"(decltype(KWc)::value));
        half4 cv = {(_Float16)fin[rr][4 * q], (_Float16)fin[rr][4 * q + 1], (_Float16)fin[rr][4 * q + 2], (_Float16)fin[rr][4 * q + 3]};
        if constexpr (RES != 0) {
            const half4 rv = rsv[j];
#pragma unroll
            for (int e = 0; e < 4; ++e)
                cv[e] = RES == 1 ? g_axpby1((float)rv[e], a.ca, (float)cv[e], a.cb) : g_axpby1((float)cv[e], a.ca, (float)rv[e], a.cb);
        }
        if constexpr (RES2 != 0) {
            const half4 rv = rsw[j];
#pragma unroll
            for (int e = 0; e < 4; ++e)
                cv[e] = RES2 == 1 ? g_axpby1((float)rv[e], a.ca2, (float)cv[e], a.cb2) : g_axpby1((float)cv[e], a.ca2, (float)rv[e], a.cb2);
        }
        bool in;
        const size_t pos = piece_pos(j, c0, y0, y1, b, &in);
        _Float16* const dst = in ? pout + pos * a.out_stride + a.out_coff + chq + 8 * q : sink;
        asm volatile("global_store_dwordx2 %0, %1, off" ::"v"(dst), "v"(cv) : "memory");
    };
    auto kloop = [&](f32x16 (&acc)[SW_R], const int b, auto&& side) {
        unsigned rowb[NIR];
#pragma unroll
        for (int ir = 0; ir < NIR; ++ir) rowb[ir] = (unsigned)((SW_R * b + ir) % SW_SLOTS) * ROWB;
        {
            const float bsel = kh ? 0.f : 1.f;      // the bias belongs to one half of the sum
            f32x16 bs;
#pragma unroll
            for (int q = 0; q < 4; ++q) {
                const f32x4 b4 = *(const f32x4*)(lbias + chq + 8 * q);
#pragma unroll
                for (int e = 0; e < 4; ++e) bs[4 * q + e] = b4[e] * bsel;
            }
#pragma unroll
            for (int r = 0; r < SW_R; ++r) acc[r] = bs;
        }
        constexpr int D = UVA_SW_D;
        half8 bq[D + 1][2];
        auto rd = [&](int idx, half8 (&dst)[2]) {      // idx = (c * 3 + dx) * NIR + ir
            const int cd = idx / NIR, ir = idx - cd * NIR, c = cd / 3, dx = cd - 3 * c;
#pragma unroll
            for (int h = 0; h < 2; ++h) dst[h] = *(const half8*)(ring + rowb[ir] + offr[dx][h] + c * (RC * 64));
        };
        __builtin_amdgcn_sched_barrier(0);
#pragma unroll
        for (int i = 0; i < D; ++i) rd(i, bq[i]);
        __builtin_amdgcn_sched_group_barrier(0x100, D * 2, 0);
        static_for<NSTEP>([&](auto I) {
            constexpr int idx = decltype(I)::value;
            constexpr int cd = idx / NIR, ir = idx - cd * NIR, c = cd / 3, dx = cd - 3 * c;
            if constexpr (idx + D < NSTEP) rd(idx + D, bq[(idx + D) % (D + 1)]);
            constexpr int ndy = (ir < 3 ? ir + 1 : 3) - (ir >= SW_R ? ir - SW_R + 1 : 0);
#pragma unroll
            for (int dy = 0; dy < 3; ++dy) {
                const int r = ir - dy;
                if (r < 0 || r >= SW_R) continue;
#pragma unroll
                for (int h = 0; h < 2; ++h)
                    acc[(r + 2 * kh) % SW_R] = __builtin_amdgcn_mfma_f32_32x32x16_f16(wgt[dy * 3 + dx][c][h], bq[idx % (D + 1)][h], acc[(r + 2 * kh) % SW_R], 0, 0, 0);
            }
            if constexpr (idx + D < NSTEP) __builtin_amdgcn_sched_group_barrier(0x100, 2, 0);
            __builtin_amdgcn_sched_group_barrier(0x008, ndy * 2, 0);
            if constexpr (idx >= E0 && (idx - E0) % ES == 0 && (idx - E0) / ES < NT) side(std::integral_constant<int, (idx - E0) / ES>{});
        });
        __builtin_amdgcn_sched_barrier(0);
    };
    // the pair's reduction: this wave's partials of the partner's rows -> LDS; after the barrier the partner's partials of
    // this wave's rows come back and are added.  A second barrier keeps the next block's hand-over off buffers that are
    // still being read.
    char* const my_hand = hand + wave * 8192 + lane * 16;
    const char* const partner_hand = hand + (wave ^ 2) * 8192 + lane * 16;
    auto hand_over = [&](const f32x16 (&acc)[SW_R]) {           // acc[2], acc[3]: the partner's rows
#pragma unroll
        for (int rr = 0; rr < 2; ++rr)
#pragma unroll
            for (int q = 0; q < 4; ++q)
                *(f32x4*)(my_hand + (rr * 4 + q) * 1024) = f32x4{acc[2 + rr][4 * q], acc[2 + rr][4 * q + 1], acc[2 + rr][4 * q + 2], acc[2 + rr][4 * q + 3]};
    };
    auto take_over = [&](const f32x16 (&acc)[SW_R], f32x16 (&fin)[2]) {
#pragma unroll
        for (int rr = 0; rr < 2; ++rr)
#pragma unroll
            for (int q = 0; q < 4; ++q) {        // (bias + first half) + second half: fp32 addition commutes
                const f32x4 got = *(const f32x4*)(partner_hand + (rr * 4 + q) * 1024);
#pragma unroll
                for (int e = 0; e < 4; ++e) fin[rr][4 * q + e] = acc[rr][4 * q + e] + got[e];
            }
        asm volatile("s_waitcnt lgkmcnt(0)\n\ts_barrier" ::: "memory");
    };

    for (int si = sb; si < se; ++si) {
        const GSwSeg seg = a.segs[si];
        const int c0 = __builtin_amdgcn_readfirstlane(seg.c0), y0 = __builtin_amdgcn_readfirstlane(seg.y0),
                  y1 = __builtin_amdgcn_readfirstlane(seg.y1);
        if (si > sb) {
            sw_barrier();
            set_plane(__builtin_amdgcn_readfirstlane(seg.plane));
            for (int rr = 0; rr < NIR; ++rr) dma_row(c0, y0, rr);
        }
        sw_barrier();
        const int nblk = (y1 - y0 + SW_R - 1) / SW_R;
        f32x16 acc[SW_R], fin[2];
        // block 0: nothing to finish yet; its sums' operands are fetched up front
        for (int rr = 0; rr < SW_R; ++rr) dma_row(c0, y0, SW_R + 2 + rr);
#pragma unroll
        for (int j = 0; j < NT; ++j) {         // (with a store to the sink in front of each: a regular block's sequence)
            if constexpr (NRL != 0) asm volatile("global_store_dwordx2 %0, %1, off" ::"v"(sink), "v"(half4{0, 0, 0, 0}) : "memory");
            load_res(j, c0, y0, y1, 0);
        }
        kloop(acc, 0, [](auto) {});
        hand_over(acc);
        asm volatile("s_waitcnt vmcnt(%0) lgkmcnt(0)\n\ts_barrier" ::"n"(NLD ? NT + NLD : 0) : "memory");
        take_over(acc, fin);
        for (int b = 1; b < nblk; ++b) {
            // the DMA of block b+1's rows, the k-loop of block b, the epilogue of block b-1 (its pieces' other operands
            // replaced by block b's as they are used): exactly NT stores and NLD loads follow the DMA pieces, so "all but
            // the newest NT + NLD memory operations have completed" proves the DMA and leaves them in flight
            if (!(UVA_SK_DBG & 4))
                for (int rr = 0; rr < SW_R; ++rr) dma_row(c0, y0, SW_R * (b + 1) + 2 + rr);
            kloop(acc, b, [&](auto J) {
                if (UVA_SK_DBG & 1) return;
                epi_piece(J, std::integral_constant<int, KW_LOOP>{}, fin, c0, y0, y1, b - 1);
                load_res(decltype(J)::value, c0, y0, y1, b);
            });
            if (!(UVA_SK_DBG & 2)) hand_over(acc);
            asm volatile("s_waitcnt vmcnt(%0) lgkmcnt(0)\n\ts_barrier" ::"n"(UVA_SK_DBG ? 0 : NT + NLD) : "memory");
            if (!(UVA_SK_DBG & 2)) take_over(acc, fin);
            else {
#pragma unroll
                for (int rr = 0; rr < 2; ++rr) fin[rr] = acc[rr] + acc[2 + rr];
            }
        }
        // the last block's pieces: behind the load of piece j's operands came the later pieces of that k-loop and now the
        // stores of the pieces before j
        static_for<NT>([&](auto J) {
            constexpr int j = decltype(J)::value;
            epi_piece(J, std::integral_constant<int, (NT - 1 - j) * (1 + NRL) + j>{}, fin, c0, y0, y1, nblk - 1);
        });
    }
    };
    if (khw) run(std::integral_constant<int, 1>{});
    else run(std::integral_constant<int, 0>{});
}

// ---------------------------------------------------------------------------------------------------------------
// rdb4_kernel: the first FOUR convolutions of a residual dense block (models/4x_Valar_v1.param:6-19),
//     x1 = lrelu(conv3(x))            x2 = lrelu(conv3(x,x1)) + conv1(x)
//     x3 = lrelu(conv3(x,x1,x2))      x4 = lrelu(conv3(x,x1,x2,x3)) + x2
// in ONE launch: x (64 channels) is read from the block's shared 192-channel array once, x1..x4 (32 channels each) are
// written into their channel ranges of the same array once -- the growing prefix never goes back to HBM between the
// convolutions (the layer-by-layer executor reads 64 + 96 + 128 + 160 channels plus halos for them).  The block's last
// convolution (192 -> 64) follows as g_conv3_sw<6, 1>.
//
// One persistent 4-wave workgroup per CU walks down a strip of 48 computed columns, one row per step, one workgroup
// barrier per step; x and the results live in row rings in LDS (64-byte chunk pixels, the XOR swizzle of g_conv3_sw).
// The four convolutions are 18 + 29 + 36 + 45 = 128 k-steps (32 input channels x one tap x 32 output channels) per row:
// 32 per wave, their weights (256 registers) stationary, every B fragment feeding two MFMAs.  A convolution that is
// split between two waves hands its partial sums over through LDS (fp32, one step later); a wave's own parts run in
// program order, so a stage may read the row its predecessor has just written:
//
//     wave 0   conv1 (18)         row s     -> x1 ring, HBM      | conv2 k-steps  0..17 (x only)     row s-1 -> P2
//     wave 1   conv2 18..26 + 1x1 row s-2   -> x2 ring, HBM      | conv3 k-steps  0..17 (x only)     row s-3 -> P3
//     wave 2   conv3 18..35       row s-4   -> x3 ring, HBM      | conv4 k-steps  0..14 (x only)     row s-5 -> P4
//     wave 3   conv4 15..44 + x2  row s-6   -> HBM               | the LDS-DMA of x row s+2
// A wave's first part's epilogue (LeakyReLU, fp16, masks, the ring and HBM stores) is spread over its second part's
// k-loop, wave 3's over the next step's: nothing but MFMAs, their fragment reads and that epilogue's VALU work between
// two barriers.
//
// Ring depths follow: x 10 rows (s-7 .. s+2), x1 8, x2 6, x3 4; 121.6 KB + 36 KB of hand-over buffers.  What a strip
// computes correctly shrinks by one column per side and convolution (x1 on 48 columns, x4 on 42); pixels outside the plane
// are written as zero (the next convolution's padding), rows above a segment's first output row are recomputed (3 of x1, 2
// of x2, 1 of x3).  Rounding points are the layer-by-layer executor's: every convolution result -> fp16, the two sums as
// g_axpby1 of fp16 operands.
struct RdbSeg { int c0, yb, ye, own0, own1, plane, pad1, pad2; };   // of plane `plane`: computed columns [c0, c0+48); output rows [yb, ye), columns [own0, own1)

struct RdbArgs {
    _Float16* arr[GEN_MAX_PLANES];   // per plane: the dense chain's array [(h+3)][(w+2)][stride]: channels 0..63 = x (read), 64..191 = x1..x4 (written)
    int ph[GEN_MAX_PLANES], pw[GEN_MAX_PLANES];
    int stride;
    const half8* w1;              // pack_generic images (natural order), cout_pad 32: [tap][cin/32][2][64][8]
    const half8* w2;
    const half8* w2s;             // the 1x1 convolution 64 -> 32 behind x2 (no bias)
    const half8* w3;
    const half8* w4;
    const float* b1;
    const float* b2;
    const float* b3;
    const float* b4;
    float slope;
    const RdbSeg* segs;
    const int* seg_begin;
    _Float16* sink;
    unsigned long long* dbg;      // UVA_INSTRUMENT builds: workgroup 0 stamps [step][wave][4] = {step start, first part done,
                                  // at the barrier, barrier passed} of its first segment here (s_memtime ticks)
};

constexpr int RA_C = 48, RA_RC = 50, RA_NF = 3;
constexpr int RA_CHB = RA_RC * 64;                                   // one chunk row: 3200 bytes
constexpr int RA_XS = 10, RA_1S = 8, RA_2S = 6, RA_3S = 4;           // ring rows
constexpr int RA_X_OFF = 0, RA_X1_OFF = RA_X_OFF + RA_XS * 2 * RA_CHB, RA_X2_OFF = RA_X1_OFF + RA_1S * RA_CHB,
              RA_X3_OFF = RA_X2_OFF + RA_2S * RA_CHB, RA_P_OFF = RA_X3_OFF + RA_3S * RA_CHB;
// rdb4_kernel's rings use g_conv3_sw's row layout with a FOUR-valued slot swizzle, unit u of ring column rc in slot
// u ^ ((rc >> 1) & 3): as conflict-free for the ds_read_b128 fragment reads as g_conv3_sw's two-valued one (the same brute
// force over all origins), and the result rows' ds_write_b64 (16 contiguous lanes = 16 pixels, the same 8 bytes of each)
// then land 2 to a bank pair instead of 4 -- within the cycles the store takes anyway (MI355X_MICROARCH.md, LDS table);
// with the two-valued swizzle 10.8 % of the kernel's LDS cycles were conflict cycles (profiles/r03_b_valar_pmc.txt).
__device__ __host__ constexpr int ra_swz(int rc) { return (rc >> 1) & 3; }
constexpr int RA_PB = RA_NF * 2 * 1024;                              // one hand-over buffer: 6 accumulator tiles
constexpr int RA_PRM_OFF = RA_P_OFF + 3 * 2 * RA_PB;
constexpr int rdb4_lds_bytes() { return RA_PRM_OFF + 4 * 32 * 4; }
static_assert(rdb4_lds_bytes() <= 160 * 1024, "rdb4 kernel LDS budget");
constexpr int RA_LAG = 9;                                            // steps of a segment beyond its output rows
constexpr int RA_NDMA = (2 * RA_RC * 4 + 63) / 64;                   // LDS-DMA pieces of an x row (the last one 16 lanes)

__global__ __launch_bounds__(256, 1) void rdb4_kernel(RdbArgs a)
{
    extern __shared__ __attribute__((aligned(16))) char smem[];
    const int wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6), lane = threadIdx.x & 63;
    const int o = lane >> 4, p = lane & 15;
    float* const prm = (float*)(smem + RA_PRM_OFF);
    for (int i = threadIdx.x; i < RA_PRM_OFF / 16; i += 256) ((uint4*)smem)[i] = make_uint4(0, 0, 0, 0);
    if (threadIdx.x < 128) {
        const int k = threadIdx.x >> 5, c = threadIdx.x & 31;
        prm[threadIdx.x] = (k == 0 ? a.b1 : k == 1 ? a.b2 : k == 2 ? a.b3 : a.b4)[c];
    }
    // per-lane LDS offsets: B fragments (three tap columns) and this lane's 8 bytes of a result pixel (channel block m).
    // Recomputed from the lane id at every step (lane_offsets(opaque(lane))): as loop invariants they are hoisted and,
    // with the weights filling the register file, spilled to scratch -- whose reloads would wait for the stores in flight.
    unsigned offdx[3], wroff[2];
    auto lane_offsets = [&](const int ln) {
        const int oo = ln >> 4, pp = ln & 15;
#pragma unroll
        for (int dx = 0; dx < 3; ++dx) {
            const int rc = pp + dx;
            offdx[dx] = (unsigned)(rc * 64 + ((oo ^ ra_swz(rc)) * 16));
        }
#pragma unroll
        for (int m = 0; m < 2; ++m) {
            const int rc = pp + 1, u = 2 * m + (oo >> 1);
            wroff[m] = (unsigned)(rc * 64 + ((u ^ ra_swz(rc)) * 16) + (oo & 1) * 8);
        }
    };
    lane_offsets(lane);
    const unsigned smem_lds = lds_offset(smem);
    _Float16* const sink = a.sink + lane * 4;
    // the plane of the current segment (seg_setup)
    _Float16* parr = nullptr;
    int ph = 0, pw = 0;
    size_t pitch = 0;                                                // elements per array row

    // ---- building blocks -----------------------------------------------------------------------------------------
    // LDS address of ring row q (q = row - R0 + 8 >= 0) of ring g: 0 = x (two chunks), 1..3 = x1..x3
    auto rowaddr = [&](int g, int q) -> unsigned {
        return g == 0 ? RA_X_OFF + (unsigned)(q % RA_XS) * (2 * RA_CHB)
             : g == 1 ? RA_X1_OFF + (unsigned)(q % RA_1S) * RA_CHB
             : g == 2 ? RA_X2_OFF + (unsigned)(q % RA_2S) * RA_CHB
                      : RA_X3_OFF + (unsigned)(q % RA_3S) * RA_CHB;
    };
    // k-steps [K0, K1) of a convolution whose k-step k = chunk * 9 + tap reads chunk 0, 1 = x, 2.. = x1.. ; row q
    auto kpart = [&](auto K0c, auto K1c, const auto& wg, f32x4 (&acc)[RA_NF][2], const int q, auto&& side) {
        constexpr int K0 = decltype(K0c)::value, K1 = decltype(K1c)::value, N = K1 - K0, D = UVA_RA_D;
        constexpr int G0 = (K0 / 9 < 2) ? 0 : K0 / 9 - 1, G1 = ((K1 - 1) / 9 < 2) ? 0 : (K1 - 1) / 9 - 1;
        unsigned ra[4][3];
#pragma unroll
        for (int g = 0; g < 4; ++g)
#pragma unroll
            for (int dy = 0; dy < 3; ++dy) ra[g][dy] = (g >= G0 && g <= G1) ? rowaddr(g, q + dy - 1) : 0u;
        half8 bq[D + 1][RA_NF];
        auto rd = [&](auto KK, half8 (&dst)[RA_NF]) {
            constexpr int k = K0 + decltype(KK)::value, chunk = k / 9, tap = k % 9, dy = tap / 3, dx = tap % 3;
            constexpr int g = chunk < 2 ? 0 : chunk - 1, coff = chunk == 1 ? RA_CHB : 0;
#pragma unroll
            for (int f = 0; f < RA_NF; ++f) dst[f] = *(const half8*)(smem + ra[g][dy] + coff + offdx[dx] + ((UVA_RA_DBG & 16) ? 0 : f) * 1024);
        };
        __builtin_amdgcn_sched_barrier(0);                           // nothing else's LDS reads count as the pipeline's
        static_for<(D < N ? D : N)>([&](auto I) { rd(I, bq[decltype(I)::value]); });
        __builtin_amdgcn_sched_group_barrier(0x100, (D < N ? D : N) * RA_NF, 0);
        static_for<N>([&](auto I) {
            constexpr int i = decltype(I)::value;
            if constexpr (i + D < N) rd(std::integral_constant<int, i + D>{}, bq[(i + D) % (D + 1)]);
#pragma unroll
            for (int f = 0; f < RA_NF; ++f)
#pragma unroll
                for (int m = 0; m < 2; ++m)
                    acc[f][m] = __builtin_amdgcn_mfma_f32_16x16x32_f16(wg[i][m], bq[i % (D + 1)][f], acc[f][m], 0, 0, 0);
            if constexpr (i + D < N) __builtin_amdgcn_sched_group_barrier(0x100, RA_NF, 0);
            __builtin_amdgcn_sched_group_barrier(0x008, RA_NF * 2, 0);
            // the previous part's epilogue, one tile behind every second k-step: VALU, LDS writes and stores that the
            // scheduler may place between this part's MFMAs (no LDS READS in there: they would count as the pipeline's)
            if constexpr ((i & 1) && i / 2 < RA_NF * 2) side(std::integral_constant<int, i / 2>{});
        });
        __builtin_amdgcn_sched_barrier(0);
    };
    auto no_side = [](auto) {};
    auto acc_bias = [&](f32x4 (&acc)[RA_NF][2], int conv) {
#pragma unroll
        for (int m = 0; m < 2; ++m) {
            const f32x4 bs = *(const f32x4*)(prm + conv * 32 + 16 * m + 4 * o);
#pragma unroll
            for (int f = 0; f < RA_NF; ++f) acc[f][m] = bs;
        }
    };
    auto acc_load = [&](f32x4 (&acc)[RA_NF][2], int j, int buf) {
        if (UVA_RA_DBG & 8) { acc_bias(acc, j); return; }
        const char* const pb = smem + RA_P_OFF + (j * 2 + buf) * RA_PB + lane * 16;
#pragma unroll
        for (int f = 0; f < RA_NF; ++f)
#pragma unroll
            for (int m = 0; m < 2; ++m) acc[f][m] = *(const f32x4*)(pb + (f * 2 + m) * 1024);
    };
    auto acc_store = [&](const f32x4 (&acc)[RA_NF][2], int j, int buf) {
        if (UVA_RA_DBG & 8) {
#pragma unroll
            for (int f = 0; f < RA_NF; ++f)
#pragma unroll
                for (int m = 0; m < 2; ++m) asm volatile("" ::"v"(acc[f][m]));
            return;
        }
        char* const pb = smem + RA_P_OFF + (j * 2 + buf) * RA_PB + lane * 16;
#pragma unroll
        for (int f = 0; f < RA_NF; ++f)
#pragma unroll
            for (int m = 0; m < 2; ++m) *(f32x4*)(pb + (f * 2 + m) * 1024) = acc[f][m];
    };
    // LeakyReLU with a slope in (0, 1) (the host checks) is max(v, slope * v); then fp16
    auto lrelu16 = [&](const f32x4 v) -> half4 {
        half4 r;
#pragma unroll
        for (int j = 0; j < 4; ++j) r[j] = (_Float16)__builtin_fmaxf(v[j], v[j] * a.slope);
        return r;
    };
    auto weights = [&](auto K0c, auto K1c, auto NCHc, const half8* wpk, auto& wg) {
        constexpr int K0 = decltype(K0c)::value, K1 = decltype(K1c)::value, NCH = decltype(NCHc)::value;
        static_for<K1 - K0>([&](auto I) {
            constexpr int k = K0 + decltype(I)::value, chunk = k / 9, tap = k % 9;
#pragma unroll
            for (int m = 0; m < 2; ++m) wg[decltype(I)::value][m] = wpk[((size_t)(tap * NCH + chunk) * 2 + m) * 64 + lane];
        });
    };
    using std::integral_constant;
#define IC(n) integral_constant<int, (n)>{}

    const int sb = a.seg_begin[blockIdx.x], se = a.seg_begin[blockIdx.x + 1];
    __syncthreads();

    // Per segment and lane: which of its three pixels (fragment f) are inside the plane (bit f) and the segment's own
    // (bit 4 + f), and where its 8 bytes of channel block m go in the array's row (without the row and fragment terms).
    unsigned pmask = 0;
    _Float16* gptr[2];
    auto seg_setup = [&](const RdbSeg& sg) {
        const int pl = __builtin_amdgcn_readfirstlane(sg.plane);
        parr = a.arr[pl]; ph = a.ph[pl]; pw = a.pw[pl];
        pitch = (size_t)(pw + 2) * a.stride;
        pmask = 0;
#pragma unroll
        for (int f = 0; f < RA_NF; ++f) {
            const int x = sg.c0 + 16 * f + p;
            pmask |= (x < pw ? 1u : 0u) << f;
            pmask |= (x < pw && x >= sg.own0 && x < sg.own1 ? 1u : 0u) << (4 + f);
        }
#pragma unroll
        for (int m = 0; m < 2; ++m) gptr[m] = parr + (size_t)(sg.c0 + p + 1) * a.stride + 64 + 16 * m + 4 * o;
    };
    struct RowOut { unsigned rbase; size_t goff; unsigned keep; };     // keep: pmask's bits that count for this row
    // result row `row` (ring row q) of convolution `conv` (1..4)
    auto row_out = [&](const int conv, const int row, const int q, const RdbSeg& sg) -> RowOut {
        RowOut r;
        r.rbase = conv == 1 ? rowaddr(1, q) : conv == 2 ? rowaddr(2, q) : rowaddr(3, q);
        r.goff = (size_t)(row + 1) * pitch + 32 * (conv - 1);
        r.keep = (((row >= 0) & (row < ph)) ? 0x0fu : 0u) | (((row >= sg.yb) & (row < sg.ye)) ? 0xf0u : 0u);
        return r;
    };
    // one tile (fragment f, channel block m) of a result row: zero outside the plane -> the ring (conv < 4) and, for the
    // segment's own rows and columns, the array.  Always exactly one store (lanes that own nothing -> the sink).
    auto emit = [&](const half4 v16, auto Jc, const int conv, const RowOut& r) {
        constexpr int j = decltype(Jc)::value, f = j >> 1, m = j & 1;
        if (UVA_RA_DBG & 2) { asm volatile("" ::"v"(v16)); return; }
        const unsigned pm = pmask & r.keep;
        const bool in = (pm >> f) & 1, own = (pm >> (4 + f)) & 1;
        const uint2 raw = __builtin_bit_cast(uint2, v16);
        const half4 v = __builtin_bit_cast(half4, make_uint2(in ? raw.x : 0u, in ? raw.y : 0u));
        if (conv < 4) *(half4*)(smem + r.rbase + wroff[m] + f * 1024) = v;
        _Float16* const dst = own ? gptr[m] + r.goff + (size_t)(16 * f) * a.stride : sink;
        if (!(UVA_RA_DBG & 1)) asm volatile("global_store_dwordx2 %0, %1, off" ::"v"(dst), "v"(v) : "memory");
    };

    // x row `row` -> ring row q, by LDS-DMA (wave 3 only): rows outside [-1, h] come from the array's zero border row
    unsigned voff[RA_NDMA];
    auto dma_setup = [&](const RdbSeg& sg) {
#pragma unroll
        for (int k = 0; k < RA_NDMA; ++k) {
            const int idx = min(k * 64 + lane, 2 * RA_RC * 4 - 1);
            const int ch = idx / (RA_RC * 4), rem = idx - ch * (RA_RC * 4), rc = rem >> 2, sl = rem & 3;
            const int u = sl ^ ra_swz(rc);
            voff[k] = (unsigned)(min(sg.c0 + rc, pw + 1) * a.stride * 2 + ch * 64 + u * 16);
        }
    };
    auto dma_x = [&](int row, int R0) {
        const int ay = (row < -1 || row > ph) ? 0 : row + 1;
        const char* const src = (const char*)(parr + (size_t)ay * pitch);
        const unsigned dst = smem_lds + rowaddr(0, row - R0 + 8);
#pragma unroll
        for (int k = 0; k < RA_NDMA; ++k) {
            if (k + 1 < RA_NDMA) glds16_s(src, voff[k], dst + k * 1024);
            else if (lane < 2 * RA_RC * 4 - (RA_NDMA - 1) * 64) glds16_s(src, voff[k], dst + k * 1024);
        }
    };
    // Every wave runs the same sequence of barriers: two per segment, then one per step.  The segment loop sits INSIDE a
    // wave's role so that its weights are fetched once per launch and stay where they are.
#ifdef UVA_INSTRUMENT
#define RA_STAMP(k) do { if (a.dbg && blockIdx.x == 0 && lane == 0 && si == sb && s < 1024) a.dbg[(s * 4 + wave) * 4 + (k)] = UVA_MEMTIME(); } while (0)
#else
#define RA_STAMP(k) do { } while (0)
#endif
#define RA_STEP_BARRIER() asm volatile("s_waitcnt vmcnt(%0) lgkmcnt(0)\n\ts_barrier" ::"n"((UVA_RA_DBG & 3) ? 0 : 2 * RA_NF) : "memory")
    // k-step ranges: conv1 [0,18) and conv2 [0,18) on wave 0; conv2 [18,27) + the 1x1 and conv3 [0,18) on wave 1; conv3
    // [18,36) and conv4 [0,15) on wave 2; conv4 [15,45) on wave 3 -- 36, 29, 33 and 30 k-steps, balanced by the in-kernel
    // stamps (tools/rdb4_anatomy.py): wave 1's epilogue is the heaviest (two accumulator sets, the sum), wave 3 also
    // issues the DMA.
    if (wave == 0) {
        half8 wa[18][2], wb[18][2];
        weights(IC(0), IC(18), IC(2), a.w1, wa);
        weights(IC(0), IC(18), IC(3), a.w2, wb);
        for (int si = sb; si < se; ++si) {
            const RdbSeg sg = a.segs[si];
            const int R0 = sg.yb - 3, nsteps = sg.ye - sg.yb + RA_LAG;      // conv1's row at step 0; ring row index q = row - R0 + 8
            seg_setup(sg);
            sw_barrier();
            sw_barrier();
#pragma clang loop unroll(disable)
            for (int s = 0; s < nsteps; ++s) {
                lane_offsets(opaque(lane));
                RA_STAMP(0);
                f32x4 acc[RA_NF][2], acc2[RA_NF][2];
                acc_bias(acc, 0);
                acc_bias(acc2, 1);
                kpart(IC(0), IC(18), wa, acc, s + 8, no_side);
                RA_STAMP(1);
                const RowOut ro = row_out(1, R0 + s, s + 8, sg);
                kpart(IC(0), IC(18), wb, acc2, s + 7, [&](auto J) {
                    constexpr int j = decltype(J)::value;
                    emit(lrelu16(acc[j >> 1][j & 1]), J, 1, ro);
                });
                acc_store(acc2, 0, s & 1);
                RA_STAMP(2);
                RA_STEP_BARRIER();
                RA_STAMP(3);
            }
        }
    } else if (wave == 1) {
        half8 wa[9][2], ws[2][2], wb[18][2];
        weights(IC(18), IC(27), IC(3), a.w2, wa);
        weights(IC(0), IC(18), IC(4), a.w3, wb);
#pragma unroll
        for (int c = 0; c < 2; ++c)
#pragma unroll
            for (int m = 0; m < 2; ++m) ws[c][m] = a.w2s[((size_t)c * 2 + m) * 64 + lane];
        for (int si = sb; si < se; ++si) {
            const RdbSeg sg = a.segs[si];
            const int R0 = sg.yb - 3, nsteps = sg.ye - sg.yb + RA_LAG;
            seg_setup(sg);
            sw_barrier();
            sw_barrier();
#pragma clang loop unroll(disable)
            for (int s = 0; s < nsteps; ++s) {
                lane_offsets(opaque(lane));
                RA_STAMP(0);
                f32x4 acc[RA_NF][2], side[RA_NF][2], acc2[RA_NF][2];
                // the 1x1 convolution of x (no bias), row s-2: centre tap of both chunks
                {
                    const unsigned rx = rowaddr(0, s + 6);
#pragma unroll
                    for (int f = 0; f < RA_NF; ++f)
#pragma unroll
                        for (int m = 0; m < 2; ++m) side[f][m] = f32x4{0.f, 0.f, 0.f, 0.f};
#pragma unroll
                    for (int c = 0; c < 2; ++c)
#pragma unroll
                        for (int f = 0; f < RA_NF; ++f) {
                            const half8 b = *(const half8*)(smem + rx + c * RA_CHB + offdx[1] + f * 1024);
#pragma unroll
                            for (int m = 0; m < 2; ++m) side[f][m] = __builtin_amdgcn_mfma_f32_16x16x32_f16(ws[c][m], b, side[f][m], 0, 0, 0);
                        }
                }
                acc_load(acc, 0, (s + 1) & 1);
                acc_bias(acc2, 2);
                kpart(IC(18), IC(27), wa, acc, s + 6, no_side);
                RA_STAMP(1);
                const RowOut ro = row_out(2, R0 + s - 2, s + 6, sg);
                kpart(IC(0), IC(18), wb, acc2, s + 5, [&](auto J) {
                    constexpr int j = decltype(J)::value;
                    const half4 c3 = lrelu16(acc[j >> 1][j & 1]);
                    half4 v;
#pragma unroll
                    for (int e = 0; e < 4; ++e) v[e] = g_axpby1((float)c3[e], 1.f, (float)(_Float16)side[j >> 1][j & 1][e], 1.f);
                    emit(v, J, 2, ro);
                });
                acc_store(acc2, 1, s & 1);
                RA_STAMP(2);
                RA_STEP_BARRIER();
                RA_STAMP(3);
            }
        }
    } else if (wave == 2) {
        half8 wa[18][2], wb[15][2];
        weights(IC(18), IC(36), IC(4), a.w3, wa);
        weights(IC(0), IC(15), IC(5), a.w4, wb);
        for (int si = sb; si < se; ++si) {
            const RdbSeg sg = a.segs[si];
            const int R0 = sg.yb - 3, nsteps = sg.ye - sg.yb + RA_LAG;
            seg_setup(sg);
            sw_barrier();
            sw_barrier();
#pragma clang loop unroll(disable)
            for (int s = 0; s < nsteps; ++s) {
                lane_offsets(opaque(lane));
                RA_STAMP(0);
                f32x4 acc[RA_NF][2], acc2[RA_NF][2];
                acc_load(acc, 1, (s + 1) & 1);
                acc_bias(acc2, 3);
                kpart(IC(18), IC(36), wa, acc, s + 4, no_side);
                RA_STAMP(1);
                const RowOut ro = row_out(3, R0 + s - 4, s + 4, sg);
                kpart(IC(0), IC(15), wb, acc2, s + 3, [&](auto J) {
                    constexpr int j = decltype(J)::value;
                    emit(lrelu16(acc[j >> 1][j & 1]), J, 3, ro);
                });
                acc_store(acc2, 2, s & 1);
                RA_STAMP(2);
                RA_STEP_BARRIER();
                RA_STAMP(3);
            }
        }
    } else {
        half8 wa[30][2];
        weights(IC(15), IC(45), IC(5), a.w4, wa);
        for (int si = sb; si < se; ++si) {
            const RdbSeg sg = a.segs[si];
            const int R0 = sg.yb - 3, nsteps = sg.ye - sg.yb + RA_LAG;
            seg_setup(sg);
            dma_setup(sg);
            sw_barrier();                                           // the previous segment is done with the rings
            dma_x(R0 - 1, R0);
            dma_x(R0, R0);
            dma_x(R0 + 1, R0);
            sw_barrier();
            // This wave has one part per step and nothing of its own to hide its epilogue behind: the epilogue of the row
            // of step s-1 (LeakyReLU, + x2 of the same pixels -- BinaryOp Add_14 -- from its ring, the stores) runs inside
            // the k-loop of step s.
            f32x4 accp[RA_NF][2];
#pragma unroll
            for (int f = 0; f < RA_NF; ++f)
#pragma unroll
                for (int m = 0; m < 2; ++m) accp[f][m] = f32x4{0.f, 0.f, 0.f, 0.f};
            auto finish = [&](auto J, const half4 (&x2)[RA_NF][2], const RowOut& ro) {
                constexpr int j = decltype(J)::value;
                const half4 c4 = lrelu16(accp[j >> 1][j & 1]);
                half4 v;
#pragma unroll
                for (int e = 0; e < 4; ++e) v[e] = g_axpby1((float)c4[e], 1.f, (float)x2[j >> 1][j & 1][e], 1.f);
                emit(v, J, 4, ro);
            };
#pragma clang loop unroll(disable)
            for (int s = 0; s < nsteps; ++s) {
                lane_offsets(opaque(lane));
                RA_STAMP(0);
                if (!(UVA_RA_DBG & 4)) dma_x(R0 + s + 2, R0);
                f32x4 acc[RA_NF][2];
                // x2 of the PREVIOUS step's row (ring row s + 1; it stays in its ring until the step after this one)
                const char* const r2 = smem + rowaddr(2, s + 1);
                half4 x2[RA_NF][2];
#pragma unroll
                for (int f = 0; f < RA_NF; ++f)
#pragma unroll
                    for (int m = 0; m < 2; ++m) x2[f][m] = *(const half4*)(r2 + wroff[m] + f * 1024);
                acc_load(acc, 2, (s + 1) & 1);
                const RowOut ro = row_out(4, R0 + s - 7, s + 1, sg);
                kpart(IC(15), IC(45), wa, acc, s + 2, [&](auto J) { finish(J, x2, ro); });
                RA_STAMP(1);
#pragma unroll
                for (int f = 0; f < RA_NF; ++f)
#pragma unroll
                    for (int m = 0; m < 2; ++m) accp[f][m] = acc[f][m];
                RA_STAMP(2);
                RA_STEP_BARRIER();
                RA_STAMP(3);
            }
            {   // the last step's row
                lane_offsets(opaque(lane));
                const char* const r2 = smem + rowaddr(2, nsteps + 1);
                half4 x2[RA_NF][2];
#pragma unroll
                for (int f = 0; f < RA_NF; ++f)
#pragma unroll
                    for (int m = 0; m < 2; ++m) x2[f][m] = *(const half4*)(r2 + wroff[m] + f * 1024);
                const RowOut ro = row_out(4, R0 + nsteps - 7, nsteps + 1, sg);
                static_for<RA_NF * 2>([&](auto J) { finish(J, x2, ro); });
            }
        }
    }
#undef RA_STEP_BARRIER
#undef RA_STAMP
#undef IC
}

}  // namespace uva
