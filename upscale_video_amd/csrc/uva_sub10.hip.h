// uva_sub10.hip.h -- sub10_kernel, gfx950 only.  Compiled in a translation unit of its own (uva_sub10.hip), like uva_wino.hip and
// uva_sub5.hip: a change here costs seconds, not uva_api.hip's minutes.  (Until round 6 this block sat in uva_kernels.hip.h.)
#pragma once
#include "uva_devutil.hip.h"
#include "uva_model.h"
#include "uva_sub10.h"

#ifndef UVA_MEMTIME
#ifdef UVA_INSTRUMENT
#define UVA_MEMTIME() __builtin_amdgcn_s_memtime()
#else
#define UVA_MEMTIME() 0ull
#endif
#endif

namespace uva {

// ----------------------------------------------------------------------------------------------
// sub10_kernel: the WHOLE 1x HurrDeblur SubCompact net (conv 3->24, 8 x conv 24->24, conv 24->3, + input, u8 in ->
// u8 out; models/1x_HurrDeblur_SubCompact_nf24-nc8_244k_net_g.param:3-26) in one launch: only the u8 frame touches
// HBM (6 B per pixel instead of ~870).
//
// One 10-wave workgroup per CU is a systolic pipeline: WAVE s IS LAYER s.  Its weights (<= 56 registers) never
// move; rows of an 80-column strip stream top to bottom through 4-row rings in LDS, one ring per layer output
// (48 B per pixel), wave s reading rows r-1..r+1 of ring s-1 and writing row r of ring s, two rows behind wave s-1;
// one workgroup barrier per row.  What is correct shrinks by one column per side and layer, so 60 of a strip's 80 columns
// are valid in the last layer (the same happens at the top of a strip segment: it starts 10 rows early); a layer is
// computed where a stored pixel needs it -- layers 1..6 on all five 16-column fragments, layers 7, 8 and the last on four
// (S10_BAL), on the rows within reach of a stored row (S10_ROWSKIP).  Pixels outside the plane are written as zero by every layer (the next layer's
// zero padding).  The head wave also loads the u8 rows (as [B,G,R,0] fp16, the 1/255 goes to the fp32 accumulator);
// the tail wave adds the input pixel, *255, rounds half-even, saturates and stores 3 bytes per pixel.
// ----------------------------------------------------------------------------------------------
constexpr int S10_ROWPX = S10_WC + 2;            // ring row: one margin pixel either side (never written)
constexpr int S10_PIXB = 48;
constexpr int S10_ROWB = S10_ROWPX * S10_PIXB;
constexpr int S10_RINGB = 4 * S10_ROWB;
constexpr int S10_UROWB = S10_ROWPX * 8;         // input ring: [B, G, R, 0] fp16 per pixel
constexpr int S10_URINGB = 4 * S10_UROWB;
constexpr int S10_PRMB = S10_NL * 96 * 4;        // per layer: bias[32], slope[32], spare[32]
constexpr int S10_RES_ROWS = 32;                 // u8 input rows kept for the residual add: the last layer writes 29 rows behind
constexpr int S10_RESB = S10_RES_ROWS * S10_ROWPX * 4;
#define S10_ROWSKIP 1                            // a layer's wave skips the rows nobody reads (round 5, block 20); 0: every layer computes every row
constexpr int S10_SKIP_ALL = 15 * 2;             // descriptor word of "no row": distance 15, which no layer takes
__host__ __device__ constexpr int sub10_lag(int stage) { return 2 * stage + 2; }
constexpr int sub10_lds_bytes() { return (S10_NL - 1) * S10_RINGB + S10_URINGB + S10_PRMB + S10_RESB + S10_MAX_ROWS * 8; }
static_assert(sub10_lds_bytes() <= 160 * 1024, "sub10 kernel LDS budget");

struct Sub10Lds {
    char* smem;
    char* uring;
    float* prm;
    char* resring;
    int2* rows;
};
__device__ __forceinline__ Sub10Lds sub10_lds(char* smem)
{
    Sub10Lds l;
    l.smem = smem;
    l.uring = smem + (S10_NL - 1) * S10_RINGB;
    l.prm = (float*)(l.uring + S10_URINGB);
    l.resring = (char*)l.prm + S10_PRMB;
    l.rows = (int2*)(l.resring + S10_RESB);
    return l;
}
// a descriptor's row word (>> 5): the frame of the batch above bit S10_FSHIFT, the plane row + S10_YBIAS below
__device__ __forceinline__ int sub10_row_y(int yb) { return (yb & ((1 << S10_FSHIFT) - 1)) - S10_YBIAS; }
__device__ __forceinline__ int sub10_row_frame(int yb) { return (yb >> S10_FSHIFT) & (S10_MAXB - 1); }
__device__ __forceinline__ void sub10_barrier() { asm volatile("s_waitcnt lgkmcnt(0)\n\ts_barrier" ::: "memory"); }

#ifdef UVA_INSTRUMENT
#define S10_STAMP(k) do { if (stamp) a.dbg[(t * S10_NW + wave) * 4 + (k)] = UVA_MEMTIME(); } while (0)
#else
#define S10_STAMP(k) do { } while (0)
#endif

// per-lane parameters of a 24-channel layer's epilogue: block 0 rows 4o..4o+3 = channels 4o..4o+3, block 1 rows
// 4o, 4o+1 = channels 16+2o, 16+2o+1 (pack_sub16)
struct Sub10Prm {
    half2v h[3];          // the slopes as packed halves
};
__device__ __forceinline__ Sub10Prm sub10_params(const float* myprm, int o)
{
    Sub10Prm q;
    const f32x4 s0 = *(const f32x4*)(myprm + 32 + 4 * o);
    const f32x2 s1 = *(const f32x2*)(myprm + 32 + 16 + 2 * o);
    q.h[0] = half2v{(_Float16)s0[0], (_Float16)s0[1]};
    q.h[1] = half2v{(_Float16)s0[2], (_Float16)s0[3]};
    q.h[2] = half2v{(_Float16)s1[0], (_Float16)s1[1]};
    return q;
}
// PReLU (x already holds the bias) as max(x, slope*x) -- channels with a slope above 1 arrive negated, the host folded
// the sign into the weights (uva_model.h pack_sub16) -- -> fp16 -> this lane's 8 + 4 bytes of a ring pixel
template <bool MASKED>
__device__ __forceinline__ void sub10_store(const f32x4 x0, const f32x4 x1, const Sub10Prm& q, char* px0, char* px1, bool inside)
{
    const f32x2 xa = {x0[0], x0[1]}, xb = {x0[2], x0[3]}, xc = {x1[0], x1[1]};
    // on packed halves, like trunkw_kernel's TW_ACT_F16 (uva_wino.h): the sum rounded to fp16, times the fp16 slope, max of the
    // two -- nine instructions per fragment instead of twelve (-DS10_ACT_F32: the fp32 form below)
    const half2v ha = __builtin_convertvector(xa, half2v), hb = __builtin_convertvector(xb, half2v), hc = __builtin_convertvector(xc, half2v);
    uint2 w0;
    w0.x = __builtin_bit_cast(unsigned, __builtin_elementwise_max(ha, ha * q.h[0]));
    w0.y = __builtin_bit_cast(unsigned, __builtin_elementwise_max(hb, hb * q.h[1]));
    unsigned w1 = __builtin_bit_cast(unsigned, __builtin_elementwise_max(hc, hc * q.h[2]));
    if (MASKED && !inside) { w0 = make_uint2(0, 0); w1 = 0; }
    *(uint2*)px0 = w0;
    *(unsigned*)px1 = w1;
}

// ---- two waves: u8 rows in, conv 3 -> 24 (+bias, PReLU); HALF 0: ring columns 0..40, fragments 0..2; HALF 1: the rest ----
template <int HALF>
__device__ __forceinline__ void sub10_head(const Sub10Args& a, const Sub10Lds L, const int wave, const int lane, const int nrows,
                                           const int nsteps)
{
    constexpr int F0 = HALF ? 3 : 0, F1 = HALF ? 5 : 3, NF = F1 - F0;
    constexpr int Q0 = HALF ? S10_ROWPX / 2 : 0, QN = S10_ROWPX / 2;     // one ring column per lane (41 lanes)
    const int p = lane & 15, o = lane >> 4;
    const bool stamp = a.dbg != nullptr && blockIdx.x == 0 && lane == 0;
    (void)stamp;
    half8 wgt[2][2];
#pragma unroll
    for (int ks = 0; ks < 2; ++ks)
#pragma unroll
        for (int m = 0; m < 2; ++m) wgt[ks][m] = a.wpk[0][(ks * 2 + m) * 64 + lane];
    // K octet ko = 4ks + o holds taps 2ko and 2ko+1 as [B,G,R,0] each; taps past 8 meet zero weights
    int sel_lo[2], sel_hi[2];
#pragma unroll
    for (int ks = 0; ks < 2; ++ks) {
        const int ta = min(2 * (4 * ks + o), 8), tb = min(2 * (4 * ks + o) + 1, 8);
        sel_lo[ks] = ((ta / 3) << 16) | ((ta % 3) * 8 + p * 8);
        sel_hi[ks] = ((tb / 3) << 16) | ((tb % 3) * 8 + p * 8);
    }
    const Sub10Prm q = sub10_params(L.prm, o);
    const f32x4 hb0 = *(const f32x4*)(L.prm + 4 * o);
    const f32x2 hb1 = *(const f32x2*)(L.prm + 16 + 2 * o);
    const float norm = (float)(1 / 255.0);
    const f32x4 zero4 = {0.f, 0.f, 0.f, 0.f};
    // the u8 row of descriptor r (two pixels per lane: ring columns lane and lane + 64), packed B | G<<8 | R<<16;
    // outside the plane: 0.  Fetched one step before it is needed so that HBM latency has a whole step to pass.
    auto fetch_row = [&](int r, unsigned& px) {
        px = 0;
        if (r < nrows) {
            const int2 e = L.rows[r];
            const int yb = __builtin_amdgcn_readfirstlane(e.x) >> 5, x0c = e.y;
            const int y = sub10_row_y(yb);
            const int qq = Q0 + lane, X = x0c - 1 + qq;
            if (lane < QN && y >= 0 && y < a.h && X >= 0 && X < a.w) {
                const uint8_t* sp = a.src[sub10_row_frame(yb)] + (size_t)y * a.src_stride + (size_t)X * 3;
                px = (unsigned)sp[0] | ((unsigned)sp[1] << 8) | ((unsigned)sp[2] << 16);
            }
        }
    };
    auto step = [&](const int t, const unsigned upx, unsigned& upx_next) {
        S10_STAMP(0);
        fetch_row(t + 1, upx_next);
        const int d = t - 2;
        int2 e = make_int2(S10_SKIP_ALL, 0);
        if (d >= 0 && d < nrows) e = L.rows[d];
        const int ye = __builtin_amdgcn_readfirstlane(e.x), x0c = __builtin_amdgcn_readfirstlane(e.y);
        if (((ye >> 1) & 15) <= (S10_ROWSKIP ? S10_NL - 1 : 14)) {      // (rows ten away from what is written out are only fetched)
            const int y = sub10_row_y(ye >> 5);
            const bool row_in = y >= 0 && y < a.h;
            unsigned rb[3];
#pragma unroll
            for (int dy = 0; dy < 3; ++dy) rb[dy] = (unsigned)(L.uring - L.smem) + ((d + dy - 1) & 3) * S10_UROWB;
            f32x4 acc[NF][2];
#pragma unroll
            for (int f = 0; f < NF; ++f) {
                acc[f][0] = zero4; acc[f][1] = zero4;
#pragma unroll
                for (int ks = 0; ks < 2; ++ks) {
                    const int ra = sel_lo[ks] >> 16, rh = sel_hi[ks] >> 16;
                    const uint2 lo = *(const uint2*)(L.smem + (ra == 0 ? rb[0] : ra == 1 ? rb[1] : rb[2]) + (sel_lo[ks] & 0xffff) + (F0 + f) * 16 * 8);
                    const uint2 hi = *(const uint2*)(L.smem + (rh == 0 ? rb[0] : rh == 1 ? rb[1] : rb[2]) + (sel_hi[ks] & 0xffff) + (F0 + f) * 16 * 8);
                    const half8 b = __builtin_bit_cast(half8, make_uint4(lo.x, lo.y, hi.x, hi.y));
                    acc[f][0] = __builtin_amdgcn_mfma_f32_16x16x32_f16(wgt[ks][0], b, acc[f][0], 0, 0, 0);
                    acc[f][1] = __builtin_amdgcn_mfma_f32_16x16x32_f16(wgt[ks][1], b, acc[f][1], 0, 0, 0);
                }
            }
            S10_STAMP(1);
            char* const px0 = L.smem + (d & 3) * S10_ROWB + (p + 1 + 16 * F0) * S10_PIXB + 8 * o;
            char* const px1 = L.smem + (d & 3) * S10_ROWB + (p + 1 + 16 * F0) * S10_PIXB + 32 + 4 * o;
            auto epi = [&](auto masked) {
#pragma unroll
                for (int f = 0; f < NF; ++f) {
                    const int X = x0c + 16 * (F0 + f) + p;
                    const f32x4 x0 = __builtin_elementwise_fma(acc[f][0], f32x4{norm, norm, norm, norm}, hb0);
                    const f32x2 x1h = __builtin_elementwise_fma(f32x2{acc[f][1][0], acc[f][1][1]}, f32x2{norm, norm}, hb1);
                    const f32x4 x1 = {x1h[0], x1h[1], 0.f, 0.f};
                    sub10_store<decltype(masked)::value>(x0, x1, q, px0 + f * 16 * S10_PIXB, px1 + f * 16 * S10_PIXB,
                                                         row_in && X >= 0 && X < a.w);
                }
            };
            if (row_in && x0c >= 0 && x0c + S10_WC <= a.w) epi(std::false_type{});
            else epi(std::true_type{});
        }
        if (t < nrows && lane < QN) {
            // the fetched u8 pixel -> [B, G, R, 0] fp16 in ring row t (outside the plane: zeros, already in upx), and as it
            // is for the last layer's residual add
            const int qq = Q0 + lane;
            const half2v bg = {(_Float16)(float)(upx & 0xff), (_Float16)(float)((upx >> 8) & 0xff)};
            const half2v r0 = {(_Float16)(float)((upx >> 16) & 0xff), (_Float16)0.f};
            *(uint2*)(L.uring + (t & 3) * S10_UROWB + qq * 8) = make_uint2(__builtin_bit_cast(unsigned, bg), __builtin_bit_cast(unsigned, r0));
            *(unsigned*)(L.resring + ((t & (S10_RES_ROWS - 1)) * S10_ROWPX + qq) * 4) = upx;
        }
        S10_STAMP(2);
        sub10_barrier();
    };
    // two steps per trip: the row fetched during one step is converted at the end of the next
    unsigned pxa, pxb;
    fetch_row(0, pxa);
    for (int t = 0; t < nsteps; t += 2) {
        step(t, pxa, pxb);
        step(t + 1, pxb, pxa);
    }
}

// ---- eight waves: conv 24 -> 24 (+bias, PReLU);  two waves (TAIL): conv 24 -> 3, + input pixel, -> u8 ----
// LDS reads: a B fragment is one ds_read_b128 per lane (16 pixels x the K octet of the lane's group o).  The hardware
// serves such a read in groups of 8 lanes of one o and 8 lanes of o^1 ({0-3,12-15} with {20-27}, ...), and 48-byte
// pixels would make those collide.  Two choices make every read conflict-free: MFMA column p holds pixel sub10_pix(p) --
// even pixels in the lanes {0-3,12-15}, odd ones in {4-11} -- and the two octets a k-step gives to o, o^1 differ by an
// even number of 16-byte units (uva_model.h SUB16_OCTET): one half of a group then touches even units only, the other
// odd ones.
__device__ __forceinline__ int sub10_pix(int p) { return p < 4 ? 2 * p : p >= 12 ? 2 * (p - 8) : 2 * (p - 4) + 1; }
// window row (0..2) of the octet that k-step ks gives to octet group o (octet 27 = none reads where 26 does)
__host__ __device__ constexpr int sub10_dy(int ks, int o) { return ((SUB16_OCTET[ks][o] > 26 ? 26 : SUB16_OCTET[ks][o]) / 3) / 3; }

// One row of one layer, fragments F0..F1-1.  The rings are handed over as __restrict__ pointers -- `rin` (plus the
// per-k-step offsets adr[]) is only read, `px0` / `px1` (this lane's 8 + 4 bytes of fragment 0's pixel in the output ring
// row) only written -- so that LDS reads may move above the previous fragment's LDS stores.
//
// LDS reads run seven k-steps ahead of their MFMAs: bq[ks] holds the B operand of k-step ks and is refilled for the next
// fragment as soon as it has been used (sched_group_barrier pins "two MFMAs, one read"; left to itself the scheduler
// either waits for every read right after issuing it or hoists all of them and spills).  Epilogues (PReLU, conversion,
// stores) follow one fragment behind and pile up behind the last MFMAs; the SIMD's other waves fill the matrix pipe
// meanwhile.  Measured and dropped (profiles/r02_sub10_experiments.txt): epilogue arithmetic pinned between the MFMAs,
// the first operands of the next row fetched before the barrier, a half-step phase shift between the SIMD's two trunk
// waves, 8-byte reads with swapped halves instead of the conflict-free 16-byte ones.
template <bool TAIL, int F0, int F1, int CSH, bool MASKED, int KS, int MB>
__device__ __forceinline__ void sub10_row(const char* __restrict__ rin, char* __restrict__ px0, char* __restrict__ px1,
                                          const char* __restrict__ res, uint8_t* __restrict__ dst, const unsigned (&adr)[KS],
                                          const half8 (&wgt)[KS][MB], const f32x4 (&binit)[2], const Sub10Prm& q, const int x0c,
                                          const int w, const bool row_in, const bool emit, const int pix, const int o)
{
    const float norm = (float)(1 / 255.0);
    half8 bq[KS];
#pragma unroll
    for (int ks = 0; ks < KS; ++ks) bq[ks] = *(const half8*)(rin + adr[ks] + F0 * 16 * S10_PIXB);
    __builtin_amdgcn_sched_group_barrier(0x100, KS, 0);
    auto mma = [&](const int f, f32x4 (&acc)[MB]) {
#pragma unroll
        for (int ks = 0; ks < KS; ++ks) {
#pragma unroll
            for (int m = 0; m < MB; ++m)
                acc[m] = __builtin_amdgcn_mfma_f32_16x16x32_f16(wgt[ks][m], bq[ks], ks == 0 ? binit[m] : acc[m], 0, 0, 0);
            if (f + 1 < F1) bq[ks] = *(const half8*)(rin + adr[ks] + (f + 1) * 16 * S10_PIXB);
            __builtin_amdgcn_sched_group_barrier(0x008, MB, 0);
            if (f + 1 < F1) __builtin_amdgcn_sched_group_barrier(0x100, 1, 0);
        }
    };
    auto epi = [&](const int f, const f32x4 (&acc)[MB]) {
        const int c = 16 * f + CSH + pix, X = x0c + c;          // (CSH: the caller's pointers already point CSH columns on)
        if constexpr (!TAIL) {
            // PReLU, zero outside the plane, fp16 -> this layer's ring row (ring column = computed column + 1)
            sub10_store<MASKED>(acc[0], acc[MB - 1], q, px0 + f * 16 * S10_PIXB, px1 + f * 16 * S10_PIXB, row_in && X >= 0 && X < w);
        } else {
            // + input pixel (Interp x1 = identity, BinaryOp add; left in LDS by the head waves), *255, cv2 convertTo(CV_8U);
            // only rows that are written out and only the columns this strip gets right
            // (channel j is lane group j's first result register -- pack_sub16 -- : three groups, one byte each, one store)
            if (emit && row_in && o < 3 && c >= S10_NL && c < S10_WC - S10_NL && X >= 0 && X < w) {
#pragma clang fp contract(off)
                const unsigned r8 = *(const unsigned*)(res + f * 16 * 4);
                // v_cvt_pk_u8_f32 rounds half to even and saturates: cv2's convertTo(CV_8U) in one instruction
                // (three roundings, as the oracle has them: x * (1/255), +, * 255.  Left alone hipcc contracts the first two into one
                // v_fmac_f32 here and not in sub5_kernel, where the product is formed in another block: the two kernels then differ
                // in three samples of a 300 x 700 frame)
                const float r = (float)((r8 >> (8 * o)) & 0xff) * norm;
                const float v = acc[0][0] + r;
                dst[f * 16 * 3 + o] = (uint8_t)__builtin_amdgcn_cvt_pk_u8_f32(v * 255.0f, 0, 0u);
            }
        }
    };
    f32x4 a0[MB], a1[MB];
    if constexpr (F1 - F0 == 5) {
        mma(F0, a0);
        mma(F0 + 1, a1);
        epi(F0, a0);
        mma(F0 + 2, a0);
        epi(F0 + 1, a1);
        mma(F0 + 3, a1);
        epi(F0 + 2, a0);
        mma(F0 + 4, a0);
        epi(F0 + 3, a1);
        epi(F0 + 4, a0);
    } else if constexpr (F1 - F0 == 4) {
        mma(F0, a0);
        mma(F0 + 1, a1);
        epi(F0, a0);
        mma(F0 + 2, a0);
        epi(F0 + 1, a1);
        mma(F0 + 3, a1);
        epi(F0 + 2, a0);
        epi(F0 + 3, a1);
    } else if constexpr (F1 - F0 == 3) {
        mma(F0, a0);
        mma(F0 + 1, a1);
        epi(F0, a0);
        mma(F0 + 2, a0);
        epi(F0 + 1, a1);
        epi(F0 + 2, a0);
    } else {
        static_assert(F1 - F0 == 2, "fragment counts 2..5");
        mma(F0, a0);
        mma(F0 + 1, a1);
        epi(F0, a0);
        epi(F0 + 1, a1);
    }
}

// CSH: the wave's fragments cover computed columns CSH + 16 F0 .. CSH + 16 F1 - 1 (see S10_BAL at the kernel)
template <bool TAIL, int F0, int F1, int CSH = 0>
__device__ __forceinline__ void sub10_body(const Sub10Args& a, const Sub10Lds L, const int wave, const int stage, const int lane,
                                           const int nrows, const int nsteps)
{
    constexpr int KS = 7, MB = TAIL ? 1 : 2;
    const int p = lane & 15, o = lane >> 4, pix = sub10_pix(p);
    const bool stamp = a.dbg != nullptr && blockIdx.x == 0 && lane == 0;
    (void)stamp;
    half8 wgt[KS][MB];
#pragma unroll
    for (int ks = 0; ks < KS; ++ks)
#pragma unroll
        for (int m = 0; m < MB; ++m) wgt[ks][m] = a.wpk[stage][(ks * MB + m) * 64 + lane];
    // per k-step: the LDS address this lane's K octet is read from.  They are kept for the row the wave works on next and
    // move one ring row per step.
    const int lag = sub10_lag(stage);
    const int maxdist = S10_ROWSKIP ? S10_NL - 1 - stage : 14;
    const unsigned in_ring = (unsigned)(stage - 1) * S10_RINGB;
    unsigned adr[KS];
#pragma unroll
    for (int ks = 0; ks < KS; ++ks) {
        const int oct = min((int)(o == 0 ? SUB16_OCTET[ks][0] : o == 1 ? SUB16_OCTET[ks][1] : o == 2 ? SUB16_OCTET[ks][2] : SUB16_OCTET[ks][3]), 26);
        const int tap = oct / 3;
        // the wave's first row is d = 0 (at step t = lag): tap row dy reads ring row (dy - 1) & 3
        adr[ks] = in_ring + ((tap / 3 - 1) & 3) * S10_ROWB + (tap % 3 + pix + CSH) * S10_PIXB + (oct % 3) * 16;
    }
    const float* const myprm = L.prm + stage * 96;
    const Sub10Prm q = sub10_params(myprm, o);
    const f32x4 binit[2] = {*(const f32x4*)(myprm + 4 * o), f32x4{myprm[16 + 2 * o], myprm[17 + 2 * o], 0.f, 0.f}};
    char* const out_ring = L.smem + stage * S10_RINGB;

    int2 desc = make_int2(0, 0);        // the descriptor of the next step's row, fetched a step ahead
    for (int t = 0; t < nsteps; ++t) {
        S10_STAMP(0);
        const int d = t - lag;
        if (d >= 0 && d < nrows) {
            const int ye = __builtin_amdgcn_readfirstlane(desc.x), x0c = __builtin_amdgcn_readfirstlane(desc.y);
            const int yy = sub10_row_y(ye >> 5);
            const bool row_in = yy >= 0 && yy < a.h;
            // rows further than 9 - stage from the nearest row that is written out are nobody's input (a segment's first and last
            // rows: 2 stage + 2 of them per segment); the ring row keeps what it held
            if (((ye >> 1) & 15) <= maxdist) {
            char* const px = out_ring + (d & 3) * S10_ROWB + (pix + 1 + CSH) * S10_PIXB;
            const char* const res = L.resring + ((d & (S10_RES_ROWS - 1)) * S10_ROWPX + pix + 1 + CSH) * 4;
            uint8_t* const dst = (TAIL ? a.dst[sub10_row_frame(ye >> 5)] : a.dst[0]) + (size_t)yy * a.dst_stride + (size_t)(x0c + pix + CSH) * 3;
            if (!TAIL && row_in && x0c >= 0 && x0c + S10_WC <= a.w)
                sub10_row<TAIL, F0, F1, CSH, false, KS, MB>(L.smem, px + 8 * o, px + 32 + 4 * o, res, dst, adr, wgt, binit, q, x0c, a.w,
                                                       row_in, (ye & 1) != 0, pix, o);
            else
                sub10_row<TAIL, F0, F1, CSH, true, KS, MB>(L.smem, px + 8 * o, px + 32 + 4 * o, res, dst, adr, wgt, binit, q, x0c, a.w,
                                                      row_in, (ye & 1) != 0, pix, o);
            }
            // Next row: every address one ring row on, wrapping after the fourth.  Whether an address wraps depends only
            // on the window row dy its octet comes from -- ring row (d + dy - 1) & 3 now -- so the three increments are
            // scalars; and in all but two k-steps (SUB16_OCTET) the four octet groups share one dy: one add each.
            int inc[3];
#pragma unroll
            for (int dy = 0; dy < 3; ++dy) inc[dy] = ((d + dy - 1) & 3) == 3 ? -3 * S10_ROWB : S10_ROWB;
#pragma unroll
            for (int ks = 0; ks < KS; ++ks) {
                const int d0 = sub10_dy(ks, 0), d1 = sub10_dy(ks, 1), d2 = sub10_dy(ks, 2), d3 = sub10_dy(ks, 3);
                if (d0 == d1 && d1 == d2 && d2 == d3) adr[ks] += (unsigned)inc[d0];
                else adr[ks] += (unsigned)(o == 0 ? inc[d0] : o == 1 ? inc[d1] : o == 2 ? inc[d2] : inc[d3]);
            }
        }
        S10_STAMP(2);
        if (d + 1 >= 0 && d + 1 < nrows) desc = L.rows[d + 1];
        sub10_barrier();
    }
}

__global__ __launch_bounds__(64 * S10_NW, 1) UVA_NO_PK_F32 void sub10_kernel(Sub10Args a)
{
    extern __shared__ __attribute__((aligned(16))) char smem[];
    const Sub10Lds L = sub10_lds(smem);
    const int wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
    const int lane = threadIdx.x & 63;
    const int nrows = __builtin_amdgcn_readfirstlane(a.nrows[blockIdx.x]);
    if (nrows <= 0) return;
    // this workgroup's row descriptors live in LDS, 8 bytes each: {32 ((frame << 16) + y + 16) + 2 dist + emit, x0} (dist: rows between y and the nearest row
    // of its segment that is written out, 0..10 -- layer s is needed where dist <= 9 - s); every wave reads one (or two) per step
    {
        const uint4* const grows = a.rows + (size_t)blockIdx.x * a.max_rows;
        for (int i = threadIdx.x; i < nrows; i += 64 * S10_NW) {
            const uint4 e = grows[i];
            L.rows[i] = make_int2(((int)((e.z >> 8) << S10_FSHIFT) + (int)e.x + S10_YBIAS) * 32 + (int)(e.w & 15u) * 2 + (int)(e.z & 1), (int)e.y);
        }
    }
    // rings start as zeros (margins and pipeline fill are never written: no NaN patterns may sit there)
    for (int i = threadIdx.x; i < ((S10_NL - 1) * S10_RINGB + S10_URINGB) / 16; i += 64 * S10_NW)
        ((uint4*)smem)[i] = make_uint4(0, 0, 0, 0);
    if (wave < S10_NL && lane < 32) {
        L.prm[wave * 96 + lane] = a.bias[wave][lane];
        L.prm[wave * 96 + 32 + lane] = wave + 1 < S10_NL ? a.slope[wave][lane] : 0.f;
    }
    __syncthreads();
    // every wave runs the same number of steps = barriers, whatever code it sits in
    const int nsteps = (nrows + S10_DRAIN + 1) & ~1;
    // Waves w, w+4, w+8 share a SIMD: two trunk layers and one half of the first or the last layer each -- the same
    // MFMA and VALU load on all four.  (No s_setprio: with the light waves halved, raising them or the trunk waves
    // measured 2 % slower than leaving the arbiter alone, profiles/r02_sub10_experiments.txt.)
#define S10_BAL 1
    // Round 5 (profiles/r05_ab_results.txt block 19): only columns 10..69 of the last layer are stored, so trunk layer 8 is needed
    // on columns 9..70 and layer 7 on 8..71 -- 64 columns, FOUR fragments at a column shift of 8 (an even number of 16-byte units:
    // the conflict-free read recipe holds), where every layer computed all five.  The last layer likewise: four fragments, two per
    // wave instead of three and two.  What a SIMD carries was 3 818 / 3 458 / 3 744 / 3 224 ticks per row (head 3 fragments, head 2,
    // tail 3, tail 2 beside two five-fragment trunk waves each); now the tail halves sit beside the five-fragment layers (waves
    // 8, 9) and the head halves beside the two four-fragment ones (waves 10, 11).  The rings' columns 0..7 and 72..79 of layers 7
    // and 8 stay at the zeros the kernel starts with; what reads them is never stored.
#define S10_MAP 0       // A/B builds (block 22): 1 = the first layer's three-fragment half on wave 11 instead of 10; 2 = the four-fragment
                              // layers on the OLDER waves of their SIMDs (waves 2, 3 = layers 7, 8; waves 6, 7 = layers 3, 4)
    if (wave < 6) sub10_body<false, 0, 5>(a, L, wave, wave + 1, lane, nrows, nsteps);
    else if (wave < 8) sub10_body<false, 0, 4, 8>(a, L, wave, wave + 1, lane, nrows, nsteps);
    else if (wave == 8) sub10_body<true, 0, 2, 8>(a, L, wave, S10_NL - 1, lane, nrows, nsteps);
    else if (wave == 9) sub10_body<true, 2, 4, 8>(a, L, wave, S10_NL - 1, lane, nrows, nsteps);
    else if (wave == (S10_MAP == 1 ? 11 : 10)) sub10_head<0>(a, L, wave, lane, nrows, nsteps);
    else sub10_head<1>(a, L, wave, lane, nrows, nsteps);
}

}  // namespace uva
