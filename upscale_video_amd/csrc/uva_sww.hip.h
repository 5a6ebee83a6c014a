// uva_sww.hip.h -- g_conv3_sww<RES, RES2>: a residual dense block's LAST convolution of 4x_Valar_v1 (192 -> 64, with the block's
// `x*1.0 + conv*0.2` and, in every third block, the `rrdb_in*1.0 + that*0.2` behind it; models/4x_Valar_v1.param:16-21;
// `-m r`, upscale/upscale_processing.py:913-916) as 1-D WINOGRAD F(2,3) along x, gfx950 only.  Round 6.
//
// g_conv3_sw<6, 1> (csrc/uva_rdb.hip.h) computes this convolution directly: 432 v_mfma_f32_16x16x32_f16 per wave and block of
// 4 rows x 32 columns.  These launches sit at the package's power limit with the matrix pipes half busy, so, as
// for the 64-feature trunk (trunkw_kernel), the lever is fewer MFMAs per pixel: output columns in PAIRS
// (2p, 2p + 1) from input columns d0..d3 = 2p - 1 .. 2p + 2,
//     V0 = d0 - d2, V1 = d1 + d2, V2 = d2 - d1, V3 = d1 - d3          (fp16, one rounding each: the MFMA's B operands)
//     U0 = g0, U1 = (g0 + g1 + g2) / 2, U2 = (g0 - g1 + g2) / 2, U3 = g2   (fp16, packed on the host: pack_generic_wino)
//     Mj = sum over input channels and filter rows of Uj * Vj          (MFMA, fp32; M1 starts at the bias)
//     out(2p) = (M0 + M1) + M2,  out(2p + 1) = (M1 - M2) - M3          (fp32, then the result's fp16 rounding as before)
// -- four multiplications for two columns instead of six: 288 MFMAs per wave and block.  A strip's 32 columns are ONE fragment of
// 16 pairs.  Unlike the trunk, the transformed rows do not fit the LDS (192 channels: 24 KB per row, ten rows) -- the rows stay
// RAW in the ring, exactly g_conv3_sw's ring, and a wave transforms at fragment-read time: four ds_read_b128 (d0..d3 of its K
// octet) and sixteen v_pk_add_f16 per (input row, 32-channel chunk) feed on average eight MFMAs, and a wave has its SIMD to
// itself here (one 4-wave workgroup per CU, 512 registers per wave), so the adds issue in the MFMAs' shadow.  For that the ring
// row holds a chunk's EVEN and ODD ring columns apart ([chunk][parity][17 records][4 x 16 B]: the LDS-DMA gives every lane its
// own source address, the permutation costs nothing), and lane p reads records p / p + 1 of either parity with g_conv3_sw's
// conflict-free slot swizzle.
//
// Organisation as g_conv3_sw<6, 1>: wave m owns output channels 16m .. 16m + 15 with ALL its weights resident (4 x 3 x 6 k-steps
// x 4 registers = 288), the workgroup walks down a 32-column strip in blocks of four rows, six input rows of a block in a
// 10-slot ring, the next block's four rows on their way by LDS-DMA.  The k-loop runs INPUT ROW by input row (all six chunks of a
// row, then the next row), so that output row r is complete behind input row r + 2 and its epilogue -- output transform, fp16,
// the fused sums (g_axpby1: the one spelling), two 8-byte stores per lane -- rides in the k-loop of input row r + 3; only the
// last row's epilogue runs behind the loop.  One accumulator set (64 registers) instead of g_conv3_sw's two.
// Numerics (the experiment valar_winograd_numerics.py of the test infrastructure; all 69 conv5 layers as F(2,3): 2.4e-3 of max|out| against fp32 where
// the direct form has 2.3e-3; u8 within 1 LSB).
#pragma once
#include "uva_devutil.hip.h"
#include "uva_sw.h"

namespace uva {

constexpr int SWW_KC = 6;                      // 32-channel chunks of the input
constexpr int SWW_RC = 34;                     // ring columns of a row: strip column -1 .. 32
constexpr int SWW_NREC = SWW_RC / 2;           // records per parity
constexpr int SWW_CHB = SWW_RC * 64;           // bytes of one chunk of a ring row
constexpr int SWW_NP = (SWW_KC * SWW_RC * 4 + 63) / 64;    // 1-KiB LDS-DMA pieces per ring row
constexpr int SWW_ROWB = SWW_NP * 1024;
constexpr int SWW_SLOTS = 10;
#ifndef SWW_WA
#define SWW_WA 44                               // weight k-steps (of 72, four registers each) that live in AccVGPRs
#endif
constexpr int sww_lds_bytes() { return SWW_SLOTS * SWW_ROWB + 256; }
static_assert(sww_lds_bytes() <= 160 * 1024, "g_conv3_sww LDS budget");

// RL: the FIRST sum's other operand is the convolution's own input, channels 0..63 of the same array (a dense block's
// `x*1.0 + conv5(cat(x, x1..x4))*0.2`: models/4x_Valar_v1.param:16-21) -- then it is already in the LDS ring (the centre tap's
// pixel) and is read from there, 8 bytes per lane and pixel, instead of being loaded from HBM a second time: a fifth of the
// kernel's memory traffic and every wait that came with it.  The same fp16 bytes either way.
template <int RES, int RES2, bool RL>
__global__ __launch_bounds__(256, 1) void g_conv3_sww(GSwArgs a)
{
    static_assert(!RL || RES != 0, "RL needs a first sum");
    constexpr int KC = SWW_KC, RC = SWW_RC, NREC = SWW_NREC, NP = SWW_NP, ROWB = SWW_ROWB;
    constexpr int NIR = SW_R + 2;              // input rows of a block
    constexpr int NPW = (NP + 3) / 4;          // DMA pieces per wave and ring row
    extern __shared__ __attribute__((aligned(16))) char smem[];
    char* const ring = smem;
    float* const lbias = (float*)(smem + SWW_SLOTS * ROWB);
    const int wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6), lane = threadIdx.x & 63;
    const int o = lane >> 4, p = lane & 15;

    // DMA pieces j = wave, wave + 4, ... of a ring row: unit idx = j*64 + lane -> (chunk, parity, record, slot) -> source bytes
    // from the row's first pixel (array column c0); the slot holds unit slot ^ swz(record).  Units past the row's end (padding
    // of the last piece) re-read the last one.
    unsigned voff[NPW];
#pragma unroll
    for (int k = 0; k < NPW; ++k) {
        const int idx = min((wave + 4 * k) * 64 + lane, KC * RC * 4 - 1);
        const int ch = idx / (RC * 4), rem = idx - ch * (RC * 4), par = rem / (NREC * 4), r2 = rem - par * (NREC * 4);
        const int e = r2 >> 2, sl = r2 & 3, rc = 2 * e + par;
        const int u = sl ^ (((e >> 2) & 1) << 1);
        voff[k] = (unsigned)(rc * a.in_stride * 2 + ch * 64 + u * 16);
    }
    if (threadIdx.x < 64) lbias[threadIdx.x] = a.bias ? a.bias[threadIdx.x] : 0.f;

    // per-lane LDS read offsets inside a chunk: d0 = even record p, d1 = odd record p, d2 = even record p + 1, d3 = odd p + 1
    unsigned offd[4];
#pragma unroll
    for (int k = 0; k < 4; ++k) {
        const int e = p + (k >> 1), par = k & 1;
        offd[k] = (unsigned)(par * (NREC * 64) + e * 64 + ((o ^ (((e >> 2) & 1) << 1)) * 16));
    }
    const unsigned ring_lds = lds_offset(ring);
    _Float16* const sink = a.sink + lane * 4;
    const _Float16* pin = nullptr;
    _Float16* pout = nullptr;
    const _Float16* pres = nullptr;
    const _Float16* pres2 = nullptr;
    int ph = 0, pw = 0;
    size_t in_pitch = 0;                                           // elements per array row
    auto set_plane = [&](int pl) {
        pin = a.in[pl]; pout = a.out[pl]; pres = a.res[pl]; pres2 = a.res2[pl];
        ph = a.ph[pl]; pw = a.pw[pl];
        in_pitch = (size_t)(pw + 2) * a.in_stride;
    };
    // ring row rr of a segment holds plane row y0 - 1 + rr = array row y0 + rr (rows below the bottom border: the border
    // row again -- zeros that only feed rows nobody stores) in slot rr % 10
    auto dma_row = [&](int c0, int y0, int rr) {
        const int ay = min(y0 + rr, ph + 1);
        const char* const src = (const char*)(pin + (size_t)ay * in_pitch + (size_t)c0 * a.in_stride);
        const unsigned dst = ring_lds + (unsigned)(rr % SWW_SLOTS) * ROWB + wave * 1024;
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
    // this wave's weights: k-step (tap row dy, transformed tap j, chunk) of its 16 output channels
    half8 wgt[12][KC];
#pragma unroll
    for (int t = 0; t < 12; ++t)
#pragma unroll
        for (int c = 0; c < KC; ++c) wgt[t][c] = a.wpk[((size_t)(t * KC + c) * 4 + wave) * 64 + lane];
    // 288 weight registers do not fit the 256 ArchVGPRs.  Left alone hipcc keeps them all in VGPR-class values and, out of VGPRs,
    // parks the overflow in AccVGPRs behind a v_accvgpr_read in front of EVERY use (267 extra VALU instructions per block, in a
    // loop whose VALU slots the transform needs).  An MFMA reads its A operand from either file: the first SWW_WA k-steps'
    // weights are pinned to AccVGPR-class values here, once -- in a loop of its own behind ALL the loads (pinned one by one
    // behind its own load, every weight waited for its own memory latency: 44 round trips per launch) --, and most MFMAs take
    // theirs from there.
#pragma unroll
    for (int t = 0; t < 12; ++t)
#pragma unroll
        for (int c = 0; c < KC; ++c)
            if (t * KC + c < SWW_WA) asm volatile("" : "+a"(wgt[t][c]));

    const int ch0 = 16 * wave + 4 * o;         // this lane's four output channels
    constexpr int NL = 2 * ((RES != 0 && !RL) + (RES2 != 0));      // HBM loads of the sums' other operands per output row
    f32x4 acc[SW_R][4];                        // [output row][j]
#pragma unroll
    for (int r = 0; r < SW_R; ++r)
#pragma unroll
        for (int j = 0; j < 4; ++j) acc[r][j] = f32x4{0.f, 0.f, 0.f, 0.f};
    half4 rs[SW_R][2], rw[SW_R][2];            // the sums' other operands of a row's two pixels, fetched a BLOCK ahead
    f32x4 ev[2];                               // a row's epilogue between its slices: the output transform's two pixels ...
    half4 cv[2];                               // ... and their fp16 values on the way through the sums

    // pixel (row r of block b, column 2p + q of the strip): is it a real one, where it sits in the arrays
    auto pix_pos = [&](int r, int q, int c0, int y0, int y1, int b, bool* inside) -> size_t {
        const int y = y0 + SW_R * b + r, x = c0 + 2 * p + q;
        *inside = y < y1 && x < pw;
        return ((size_t)(min(y, ph - 1) + 1) * (pw + 2) + 1 + min(x, pw - 1));      // (clamped: lanes outside read a real pixel)
    };
    auto load_res = [&](int r, int c0, int y0, int y1, int b) {
        if constexpr (RES != 0 || RES2 != 0) {
#pragma unroll
            for (int q = 0; q < 2; ++q) {
                bool in;
                const size_t pos = pix_pos(r, q, c0, y0, y1, b, &in);
                // asm loads: invisible to hipcc's s_waitcnt bookkeeping ON PURPOSE.  Its own wait in front of a use would be
                // vmcnt(k), k = the loads IT knows of behind this one -- and would drain the next block's row DMA (asm too, issued
                // in between) half a block early.  The uses wait with the exact count instead (res_wait below).
                if constexpr (RES != 0 && !RL) asm volatile("global_load_dwordx2 %0, %1, off" : "=v"(rs[r][q]) : "v"(pres + pos * a.res_stride + ch0) : "memory");
                if constexpr (RES2 != 0) asm volatile("global_load_dwordx2 %0, %1, off" : "=v"(rw[r][q]) : "v"(pres2 + pos * a.res2_stride + ch0) : "memory");
            }
        }
    };
    // The epilogue of output row r of block b in SIX SLICES, each a few instructions beside one k-step group's MFMAs: 0 the
    // output transform, 1 fp16, 2 / 3 the first sum of the row's two pixels, 4 the second sum, 5 the two stores and the loads of
    // the sums' other operands for the SAME row of the NEXT block (a whole block ahead of their use: a wave has nobody to hide
    // a memory latency behind).  `live`: false = the stores go to the sink (the deferred last row of a block that does not exist).
    auto epi_slice = [&](auto SL, const int r, const int c0, const int y0, const int y1, const int b, const bool live) {
        constexpr int sl = decltype(SL)::value;
        if constexpr (sl == 0) {
            const f32x4 m0 = acc[r][0], m1 = acc[r][1], m2 = acc[r][2], m3 = acc[r][3];
            ev[0] = (m0 + m1) + m2;
            ev[1] = (m1 - m2) - m3;
        } else if constexpr (sl == 1) {
#pragma unroll
            for (int q = 0; q < 2; ++q) cv[q] = half4{(_Float16)ev[q][0], (_Float16)ev[q][1], (_Float16)ev[q][2], (_Float16)ev[q][3]};
            if constexpr (RL) {
                // output pixel (row r of block b, column 2p + q) = ring row 4b + r + 1, ring column 2p + q + 1: q = 0 the odd record p,
                // q = 1 the even record p + 1; this lane's four channels: chunk wave >> 1, unit 2 (wave & 1) + (o >> 1), half o & 1
                const unsigned rrow = (unsigned)((SW_R * b + r + 1) % SWW_SLOTS) * ROWB + (unsigned)(wave >> 1) * SWW_CHB + (o & 1) * 8;
                const int u = 2 * (wave & 1) + (o >> 1);
#pragma unroll
                for (int q = 0; q < 2; ++q) {
                    const int e = p + q, par = 1 - q;
                    rs[r][q] = *(const half4*)(ring + rrow + par * (NREC * 64) + e * 64 + ((u ^ (((e >> 2) & 1) << 1)) * 16));
                }
            }
        } else if constexpr (sl == 2 || sl == 3) {
            if constexpr (RES != 0) {
                constexpr int q = sl - 2;
                // The row's operands were requested in slice 5 of the same row of the previous block.  Behind them, in issue order:
                // the slice-5 operations (2 stores + NL loads) of the three other rows and the >= 12 DMA pieces of this block --
                // "all but the newest 3 (2 + NL) + 12 have completed" is exactly "they are here", and waits for nothing younger.
                if constexpr (sl == 2 && NL > 0) asm volatile("s_waitcnt vmcnt(%0)" ::"n"(3 * (2 + NL) + 12) : "memory");
                if constexpr (!RL) asm volatile("" : "+v"(rs[r][0]), "+v"(rs[r][1]));
                if constexpr (RES2 != 0) asm volatile("" : "+v"(rw[r][0]), "+v"(rw[r][1]));
                const half4 rv = rs[r][q];
#pragma unroll
                for (int e = 0; e < 4; ++e)
                    cv[q][e] = RES == 1 ? g_axpby1((float)rv[e], a.ca, (float)cv[q][e], a.cb) : g_axpby1((float)cv[q][e], a.ca, (float)rv[e], a.cb);
            }
        } else if constexpr (sl == 4) {
            if constexpr (RES2 != 0) {
#pragma unroll
                for (int q = 0; q < 2; ++q) {
                    const half4 rv = rw[r][q];
#pragma unroll
                    for (int e = 0; e < 4; ++e)
                        cv[q][e] = RES2 == 1 ? g_axpby1((float)rv[e], a.ca2, (float)cv[q][e], a.cb2) : g_axpby1((float)cv[q][e], a.ca2, (float)rv[e], a.cb2);
                }
            }
        } else {
#pragma unroll
            for (int q = 0; q < 2; ++q) {
                bool in;
                const size_t pos = pix_pos(r, q, c0, y0, y1, b, &in);
                _Float16* const dst = (in && live) ? pout + pos * a.out_stride + a.out_coff + ch0 : sink;
                const half4 val = cv[q];
                asm volatile("global_store_dwordx2 %0, %1, off" ::"v"(dst), "v"(val) : "memory");
            }
            load_res(r, c0, y0, y1, b + 1);
        }
    };

    // The k-loop of block b: group g = (input row ir, chunk c), software-pipelined by hand -- a wave has its SIMD to itself and
    // issues in order, so whatever is not requested early is waited for in full: the raw fragments are read TWO groups ahead,
    // the transform runs ONE group ahead (beside the current group's MFMAs), and an epilogue slice follows every group of the
    // input rows 1 (the PREVIOUS block's last row), 3, 4 and 5 (this block's rows 0, 1, 2).
    auto kloop = [&](const int b, const int c0, const int y0, const int y1) {
        unsigned rowb[NIR];                    // input row ir of the block = ring row 4b + ir
#pragma unroll
        for (int ir = 0; ir < NIR; ++ir) rowb[ir] = (unsigned)((SW_R * b + ir) % SWW_SLOTS) * ROWB;
        const f32x4 bias4 = *(const f32x4*)(lbias + ch0);
        const f32x4 zero4 = {0.f, 0.f, 0.f, 0.f};
        constexpr int NG = NIR * KC;
        half8 d[2][4], v[2][4];
        auto rd = [&](int g, half8 (&dst)[4]) {
            const int ir = g / KC, c = g - ir * KC;
#pragma unroll
            for (int k = 0; k < 4; ++k) dst[k] = *(const half8*)(ring + rowb[ir] + c * SWW_CHB + offd[k]);
        };
        auto transform = [&](const half8 (&dd)[4], half8 (&vv)[4]) {
            vv[0] = pk_sub(dd[0], dd[2]);
            vv[1] = dd[1] + dd[2];
            vv[2] = pk_sub(dd[2], dd[1]);
            vv[3] = pk_sub(dd[1], dd[3]);
        };
        __builtin_amdgcn_sched_barrier(0);
        rd(0, d[0]);
        rd(1, d[1]);
        transform(d[0], v[0]);
        __builtin_amdgcn_sched_barrier(0);
        static_for<NG>([&](auto G) {
            constexpr int g = decltype(G)::value, ir = g / KC, c = g - ir * KC;
            if constexpr (g + 2 < NG) rd(g + 2, d[g & 1]);
            if constexpr (g + 1 < NG) transform(d[(g + 1) & 1], v[(g + 1) & 1]);
            // B-operand-major: a transformed fragment Vj feeds its (up to three) tap rows back to back.  At the power cap the order
            // matters: neighbours that share the B operand run 10 % faster than neighbours that share nothing, neighbours that share
            // the weights only 3 % (tools/mfma_operand_order_bench.hip, profiles/r06h/) -- trunkw_kernel's order, and now this one's.
#pragma unroll
            for (int j = 0; j < 4; ++j) {
#pragma unroll
                for (int dy = 0; dy < 3; ++dy) {
                    const int r = ir - dy;
                    if (r < 0 || r >= SW_R) continue;
                    const bool first = dy == 0 && c == 0;
                    acc[r][j] = __builtin_amdgcn_mfma_f32_16x16x32_f16(wgt[dy * 4 + j][c], v[g & 1][j], first ? (j == 1 ? bias4 : zero4) : acc[r][j], 0, 0, 0);
                }
            }
            if constexpr (ir == 1) epi_slice(std::integral_constant<int, c>{}, SW_R - 1, c0, y0, y1, b - 1, b > 0);
            if constexpr (ir >= 3) epi_slice(std::integral_constant<int, c>{}, ir - 3, c0, y0, y1, b, true);
            __builtin_amdgcn_sched_barrier(0);
        });
    };

    for (int si = sb; si < se; ++si) {
        const GSwSeg seg = a.segs[si];
        const int c0 = __builtin_amdgcn_readfirstlane(seg.c0), y0 = __builtin_amdgcn_readfirstlane(seg.y0),
                  y1 = __builtin_amdgcn_readfirstlane(seg.y1);
        if (si > sb) {
            asm volatile("s_waitcnt vmcnt(0) lgkmcnt(0)\n\ts_barrier" ::: "memory");      // the previous segment's last reads are done
            set_plane(__builtin_amdgcn_readfirstlane(seg.plane));
            for (int rr = 0; rr < NIR; ++rr) dma_row(c0, y0, rr);
        }
        for (int r = 0; r < SW_R; ++r) load_res(r, c0, y0, y1, 0);
        asm volatile("s_waitcnt vmcnt(0) lgkmcnt(0)\n\ts_barrier" ::: "memory");
        const int nblk = (y1 - y0 + SW_R - 1) / SW_R;
        for (int b = 0; b < nblk; ++b) {
            // the next block's four new rows (past the last block: rows nobody reads), then this block.  The last epilogue
            // slice's two stores and NL loads are the only memory operations behind everything else of the iteration: "all but
            // the newest 2 + NL have completed" proves the DMA (operations complete in issue order) and leaves those in flight.
            for (int rr = 0; rr < SW_R; ++rr) dma_row(c0, y0, SW_R * (b + 1) + 2 + rr);
            kloop(b, c0, y0, y1);
            asm volatile("s_waitcnt vmcnt(%0) lgkmcnt(0)\n\ts_barrier" ::"n"(2 + NL) : "memory");
        }
        // the segment's last block's last row
        static_for<6>([&](auto SL) { epi_slice(SL, SW_R - 1, c0, y0, y1, nblk - 1, true); });
    }
}

}  // namespace uva
