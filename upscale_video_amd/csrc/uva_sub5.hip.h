// uva_sub5.hip.h -- sub5_kernel<PART>: the 1x HurrDeblur SubCompact net (reference graph
// models/1x_HurrDeblur_SubCompact_nf24-nc8_244k_net_g.param:3-26: conv 3->24 + PReLU, 8 x (conv 24->24 + PReLU), conv 24->3,
// PixelShuffle(1), Interp(1), + input) as TWO launches of FIVE layers, gfx950 only.
//
// Why: sub10_kernel -- the whole net in one launch, csrc/uva_kernels.hip.h -- issues MFMAs of which
// 0.47 are useful: 24 of 32 MFMA rows, 60 of 80 computed columns (ten layers eat ten columns per side), n of n + 20 rows per
// segment.  Its row period is what the three waves of a SIMD issue between two barriers, so only removed work counts.  Five
// layers per launch eat five columns and five rows per side:
//
//   part 0  u8 frame -> conv 3->24 (layer 0) -> layers 1..4 -> the 24-channel image `mid` (fp16, 48 B per pixel) in HBM
//   part 1  mid -> layers 5..8 -> conv 24->3 (layer 9) + input pixel -> u8 frame
//
// Organisation: one 12-wave workgroup per CU runs TWO pipelines on two neighbouring 64-column strips (54 valid columns each):
// waves 0-3 / 4-7 are the four 24->24 layers of pipeline 0 / 1 (a wave IS a layer, its 56 weight registers never move), waves 8, 9
// the pipelines' LIGHT FRONT wave, waves 10, 11 their LIGHT BACK wave.  The launch's light LAYER (part 0: conv 3->24, part 1:
// conv 24->3 + residual -> u8) is split between the two by fragments -- part 0: front 0, 1, back 2, 3; part 1: front 0, back 1..3
// (its front wave also waits for HBM) -- and each also
// moves rows: part 0's front wave brings the u8 rows in, its back wave takes layer 4's rows out to `mid`; part 1's front wave
// brings the rows of `mid` in.  (First version: the whole light layer on one wave, the other one only moving rows -- the
// light waves set the row period, as in sub10_kernel, profiles/r05_ab_results.txt.)  Waves w, w+4, w+8 share a SIMD: two
// trunk layers and one light wave on every SIMD, as in sub10_kernel.  Rows stream top to bottom through 4-row rings in LDS
// (one ring per layer output, 48 B per pixel, sub10_kernel's layout and conflict-free read recipe), one workgroup barrier per
// row; layer s runs two rows behind layer s-1.  Segments start 5 rows early and end 5 rows late (host: build_sub5_rows).
// The arithmetic, its order and its rounding points are sub10_kernel's: the two kernels give the same bytes (tests).
#pragma once
#include "uva_devutil.hip.h"
#include "uva_model.h"
#include "uva_sub5.h"

namespace uva {
namespace s5 {

constexpr int NW = 12;
#define S5_TAIL_SPLIT 1                          // part 1: the front wave (which also brings the rows of `mid` in: ~940 ticks per row) takes
                              // fragments [0, S5_TAIL_SPLIT) of the last layer, the back wave the rest.  With 2 / 2 the front
                                                 // wave set the row period (3 840 against 2 900 ticks, profiles/r05_ab4_sub5_anatomy.txt)
constexpr int ROWPX = S5_WC + 2;                 // ring row: one margin pixel either side
constexpr int PIXB = 48;
constexpr int ROWB = ROWPX * PIXB;               // 3168 (an even number of 16-byte units: the read recipe needs that)
constexpr int RINGB = 4 * ROWB;
constexpr int UROWB = ROWPX * 8;                 // part 0's input ring: [B, G, R, 0] fp16 per pixel
constexpr int URINGB = 4 * UROWB;
constexpr int NRINGS = 5;                        // part 0: outputs of layers 0..4; part 1: the rows of `mid` + outputs of layers 5..8
constexpr int PIPEB = NRINGS * RINGB + URINGB;   // one pipeline
constexpr int PRMB = S5_NL * 96 * 4;             // per layer: bias[32], slope[32], spare[32]
constexpr int LDS_BYTES = 2 * PIPEB + PRMB + S5_MAX_ROWS * 8;
static_assert(LDS_BYTES <= 160 * 1024, "sub5 kernel LDS budget");
static_assert((ROWB / 16) % 2 == 0 && (RINGB / 16) % 2 == 0, "row and ring strides are even numbers of 16-byte units");
static_assert(S5_MIDB == PIXB, "the image between the launches is a copy of ring pixels");

struct Lds {
    char* pipe;       // this wave's pipeline: rings 0..4, then the u8 ring
    float* prm;
    const int2* rows;
};
__device__ __forceinline__ void barrier() { asm volatile("s_waitcnt lgkmcnt(0)\n\ts_barrier" ::: "memory"); }

#ifdef UVA_INSTRUMENT
#define S5_STAMP(k) do { if (stamp) a.dbg[(t * NW + wave) * 4 + (k)] = __builtin_amdgcn_s_memtime(); } while (0)
#else
#define S5_STAMP(k) do { } while (0)
#endif

// per-lane PReLU slopes of a 24-channel layer: block 0 rows 4o..4o+3 = channels 4o..4o+3, block 1 rows 4o, 4o+1 = channels
// 16+2o, 16+2o+1 (uva_model.h pack_sub16), as packed halves
struct Prm { half2v h[3]; };
__device__ __forceinline__ Prm params(const float* myprm, int o)
{
    const f32x4 s0 = *(const f32x4*)(myprm + 32 + 4 * o);
    const f32x2 s1 = *(const f32x2*)(myprm + 32 + 16 + 2 * o);
    Prm q;
    q.h[0] = half2v{(_Float16)s0[0], (_Float16)s0[1]};
    q.h[1] = half2v{(_Float16)s0[2], (_Float16)s0[3]};
    q.h[2] = half2v{(_Float16)s1[0], (_Float16)s1[1]};
    return q;
}
// PReLU (x holds the bias already) as max(x, slope x) on packed halves -- channels with a slope above 1 arrive negated, the host
// folded the sign into the weights (pack_sub16) -- -> this lane's 8 + 4 bytes of a ring pixel.  sub10_kernel's sub10_store.
template <bool MASKED>
__device__ __forceinline__ void store_px(const f32x4 x0, const f32x4 x1, const Prm& q, char* px0, char* px1, bool inside)
{
    const f32x2 xa = {x0[0], x0[1]}, xb = {x0[2], x0[3]}, xc = {x1[0], x1[1]};
    const half2v ha = __builtin_convertvector(xa, half2v), hb = __builtin_convertvector(xb, half2v), hc = __builtin_convertvector(xc, half2v);
    uint2 w0;
    w0.x = __builtin_bit_cast(unsigned, __builtin_elementwise_max(ha, ha * q.h[0]));
    w0.y = __builtin_bit_cast(unsigned, __builtin_elementwise_max(hb, hb * q.h[1]));
    unsigned w1 = __builtin_bit_cast(unsigned, __builtin_elementwise_max(hc, hc * q.h[2]));
    if (MASKED && !inside) { w0 = make_uint2(0, 0); w1 = 0; }
    *(uint2*)px0 = w0;
    *(unsigned*)px1 = w1;
}
// MFMA column p holds pixel pix(p): even pixels in the lanes {0-3, 12-15}, odd ones in {4-11} (conflict-free 16-byte reads of
// 48-byte pixels, see sub10_kernel)
__device__ __forceinline__ int pix_of(int p) { return p < 4 ? 2 * p : p >= 12 ? 2 * (p - 8) : 2 * (p - 4) + 1; }
__host__ __device__ constexpr int dy_of(int ks, int o) { return ((SUB16_OCTET[ks][o] > 26 ? 26 : SUB16_OCTET[ks][o]) / 3) / 3; }

// ---- part 0, the two light waves of a pipeline: conv 3 -> 24 (+ bias, PReLU) on fragments [F0, F1);
//      FRONT (F0 == 0): also the u8 rows in;  BACK: also layer 4's finished rows (ring 4) out to `mid` ---------------------------------
template <int F0, int F1, bool FRONT>
__device__ __forceinline__ void head(const Sub5Args& a, const Lds L, const int wave, const int lane, const int xoff, const int nrows,
                                     const int nsteps)
{
    constexpr int NF = F1 - F0;
    const int p = lane & 15, o = lane >> 4;
    [[maybe_unused]] const bool stamp = a.dbg != nullptr && blockIdx.x == 0 && lane == 0;
    char* const uring = L.pipe + NRINGS * RINGB;
    half8 wgt[2][2];
#pragma unroll
    for (int ks = 0; ks < 2; ++ks)
#pragma unroll
        for (int m = 0; m < 2; ++m) wgt[ks][m] = ((const half8*)a.wpk[0])[(ks * 2 + m) * 64 + lane];
    // K octet ko = 4ks + o holds taps 2ko and 2ko+1 as [B,G,R,0] each; taps past 8 meet zero weights
    int sel_lo[2], sel_hi[2];
#pragma unroll
    for (int ks = 0; ks < 2; ++ks) {
        const int ta = min(2 * (4 * ks + o), 8), tb = min(2 * (4 * ks + o) + 1, 8);
        sel_lo[ks] = ((ta / 3) << 16) | ((ta % 3) * 8 + p * 8);
        sel_hi[ks] = ((tb / 3) << 16) | ((tb % 3) * 8 + p * 8);
    }
    const Prm q = params(L.prm, o);
    const f32x4 hb0 = *(const f32x4*)(L.prm + 4 * o);
    const f32x2 hb1 = *(const f32x2*)(L.prm + 16 + 2 * o);
    const float norm = (float)(1 / 255.0);
    const f32x4 zero4 = {0.f, 0.f, 0.f, 0.f};
    // FRONT: the u8 row of descriptor r: ring column `lane` (and, lanes 0 and 1, ring column 64 + lane), packed B | G<<8 | R<<16;
    // outside the plane: 0.  Fetched one step before it is needed: HBM latency has a whole step to pass.
    auto fetch_row = [&](int r, unsigned (&px)[2]) {
        px[0] = px[1] = 0;
        if (r < nrows) {
            const int2 e = L.rows[r];
            const int y = e.x >> 1, x0c = e.y + xoff;
            if (y >= 0 && y < a.h) {
#pragma unroll
                for (int k = 0; k < 2; ++k) {
                    const int qq = lane + 64 * k, X = x0c - 1 + qq;
                    if (qq < ROWPX && X >= 0 && X < a.w) {
                        const uint8_t* sp = a.src + (size_t)y * a.src_stride + (size_t)X * 3;
                        px[k] = (unsigned)sp[0] | ((unsigned)sp[1] << 8) | ((unsigned)sp[2] << 16);
                    }
                }
            }
        }
    };
    // BACK: layer 4's row of descriptor d -> `mid`, the strip's 54 valid columns, 16 bytes per lane and unit
    auto store_row = [&](const int d) {
        constexpr int UNITS = S5_VALID * 3;
        if (d < 0 || d >= nrows) return;
        const int2 e = L.rows[d];
        const int ye = __builtin_amdgcn_readfirstlane(e.x), x0c = __builtin_amdgcn_readfirstlane(e.y) + xoff;
        const int y = ye >> 1;
        if (!(ye & 1) || y < 0 || y >= a.h) return;
        const char* const src = L.pipe + 4 * RINGB + (d & 3) * ROWB + (1 + S5_NL) * PIXB;
        char* const dst = a.mid + ((size_t)y * a.w + (x0c + S5_NL)) * PIXB;
#pragma unroll
        for (int k = 0; k < (UNITS + 63) / 64; ++k) {
            const int u = lane + 64 * k;
            if (u < UNITS && x0c + S5_NL + u / 3 < a.w) *(uint4*)(dst + u * 16) = *(const uint4*)(src + u * 16);
        }
    };
    auto conv_row = [&](const int d) {
        if (d < 0 || d >= nrows) return;
        const int2 e = L.rows[d];
        const int ye = __builtin_amdgcn_readfirstlane(e.x), x0c = __builtin_amdgcn_readfirstlane(e.y) + xoff;
        const int y = ye >> 1;
        const bool row_in = y >= 0 && y < a.h;
        unsigned rb[3];
#pragma unroll
        for (int dy = 0; dy < 3; ++dy) rb[dy] = (unsigned)(uring - L.pipe) + ((d + dy - 1) & 3) * UROWB;
        f32x4 acc[NF][2];
#pragma unroll
        for (int f = 0; f < NF; ++f) {
            acc[f][0] = zero4; acc[f][1] = zero4;
#pragma unroll
            for (int ks = 0; ks < 2; ++ks) {
                const int ra = sel_lo[ks] >> 16, rh = sel_hi[ks] >> 16;
                const uint2 lo = *(const uint2*)(L.pipe + (ra == 0 ? rb[0] : ra == 1 ? rb[1] : rb[2]) + (sel_lo[ks] & 0xffff) + (F0 + f) * 16 * 8);
                const uint2 hi = *(const uint2*)(L.pipe + (rh == 0 ? rb[0] : rh == 1 ? rb[1] : rb[2]) + (sel_hi[ks] & 0xffff) + (F0 + f) * 16 * 8);
                const half8 b = __builtin_bit_cast(half8, make_uint4(lo.x, lo.y, hi.x, hi.y));
                acc[f][0] = __builtin_amdgcn_mfma_f32_16x16x32_f16(wgt[ks][0], b, acc[f][0], 0, 0, 0);
                acc[f][1] = __builtin_amdgcn_mfma_f32_16x16x32_f16(wgt[ks][1], b, acc[f][1], 0, 0, 0);
            }
        }
        char* const px0 = L.pipe + (d & 3) * ROWB + (p + 1 + 16 * F0) * PIXB + 8 * o;
        char* const px1 = L.pipe + (d & 3) * ROWB + (p + 1 + 16 * F0) * PIXB + 32 + 4 * o;
        auto epi = [&](auto masked) {
#pragma unroll
            for (int f = 0; f < NF; ++f) {
                const int X = x0c + 16 * (F0 + f) + p;
                const f32x4 x0 = __builtin_elementwise_fma(acc[f][0], f32x4{norm, norm, norm, norm}, hb0);
                const f32x2 x1h = __builtin_elementwise_fma(f32x2{acc[f][1][0], acc[f][1][1]}, f32x2{norm, norm}, hb1);
                const f32x4 x1 = {x1h[0], x1h[1], 0.f, 0.f};
                store_px<decltype(masked)::value>(x0, x1, q, px0 + f * 16 * PIXB, px1 + f * 16 * PIXB, row_in && X >= 0 && X < a.w);
            }
        };
        if (row_in && x0c >= 0 && x0c + S5_WC <= a.w) epi(std::false_type{});
        else epi(std::true_type{});
    };
    auto step = [&](const int t, const unsigned (&upx)[2], unsigned (&upx_next)[2]) {
        S5_STAMP(0);
        if constexpr (FRONT) fetch_row(t + 1, upx_next);
        conv_row(t - 2);
        S5_STAMP(1);
        if constexpr (FRONT) {
            if (t < nrows) {
                // the fetched u8 pixels -> [B, G, R, 0] fp16 in ring row t (outside the plane: zeros, already in upx)
#pragma unroll
                for (int k = 0; k < 2; ++k) {
                    const int qq = lane + 64 * k;
                    if (qq < ROWPX) {
                        const half2v bg = {(_Float16)(float)(upx[k] & 0xff), (_Float16)(float)((upx[k] >> 8) & 0xff)};
                        const half2v r0 = {(_Float16)(float)((upx[k] >> 16) & 0xff), (_Float16)0.f};
                        *(uint2*)(uring + (t & 3) * UROWB + qq * 8) = make_uint2(__builtin_bit_cast(unsigned, bg), __builtin_bit_cast(unsigned, r0));
                    }
                }
            }
        } else {
            store_row(t - (2 * 4 + 2 + 1));      // one step behind layer 4 (no row below is needed)
        }
        S5_STAMP(2);
        barrier();
    };
    unsigned pxa[2] = {0, 0}, pxb[2] = {0, 0};
    if constexpr (FRONT) fetch_row(0, pxa);
    for (int t = 0; t < nsteps; t += 2) {      // two steps per trip: the row fetched during one step is converted at the end of the next
        step(t, pxa, pxb);
        step(t + 1, pxb, pxa);
    }
}

// ---- one row of one 24-input layer, four fragments (sub10_kernel's sub10_row): LDS reads run seven k-steps ahead of their MFMAs,
// epilogues follow one fragment behind.  TAIL: conv 24 -> 3 + input pixel -> u8 (the residual bytes come from HBM: r8[]) -------------
template <bool TAIL, bool MASKED, int KS, int MB, int F0, int F1>
__device__ __forceinline__ void rowf(const char* __restrict__ rin, char* __restrict__ px0, char* __restrict__ px1, uint8_t* __restrict__ dst,
                                     const unsigned (&adr)[KS], const half8 (&wgt)[KS][MB], const f32x4 (&binit)[2], const Prm& q,
                                     const unsigned (&r8)[4], const int x0c, const int w, const bool row_in, const bool emit, const int pix,
                                     const int o)
{
    const float norm = (float)(1 / 255.0);
    half8 bq[KS];
#pragma unroll
    for (int ks = 0; ks < KS; ++ks) bq[ks] = *(const half8*)(rin + adr[ks] + F0 * 16 * PIXB);
    __builtin_amdgcn_sched_group_barrier(0x100, KS, 0);
    auto mma = [&](const int f, f32x4 (&acc)[MB]) {
#pragma unroll
        for (int ks = 0; ks < KS; ++ks) {
#pragma unroll
            for (int m = 0; m < MB; ++m)
                acc[m] = __builtin_amdgcn_mfma_f32_16x16x32_f16(wgt[ks][m], bq[ks], ks == 0 ? binit[m] : acc[m], 0, 0, 0);
            if (f + 1 < F1) bq[ks] = *(const half8*)(rin + adr[ks] + (f + 1) * 16 * PIXB);
            __builtin_amdgcn_sched_group_barrier(0x008, MB, 0);
            if (f + 1 < F1) __builtin_amdgcn_sched_group_barrier(0x100, 1, 0);
        }
    };
    auto epi = [&](const int f, const f32x4 (&acc)[MB]) {
        const int c = 16 * f + pix, X = x0c + c;
        if constexpr (!TAIL) {
            store_px<MASKED>(acc[0], acc[MB - 1], q, px0 + f * 16 * PIXB, px1 + f * 16 * PIXB, row_in && X >= 0 && X < w);
        } else {
            // + input pixel (Interp x1 = identity, BinaryOp add), *255, cv2 convertTo(CV_8U) = v_cvt_pk_u8_f32 (half to even,
            // saturating); only rows that are written out and only the columns this strip gets right
            // (channel j is lane group j's first result register -- pack_sub16 -- : three groups, one byte each, one store)
            if (emit && row_in && o < 3 && c >= S5_NL && c < S5_WC - S5_NL && X >= 0 && X < w) {
#pragma clang fp contract(off)
                // (three roundings, as the oracle has them and as sub10_kernel has them: no contraction into an fma)
                const float r = (float)r8[f] * norm;
                const float v = acc[0][0] + r;
                dst[f * 16 * 3 + o] = (uint8_t)__builtin_amdgcn_cvt_pk_u8_f32(v * 255.0f, 0, 0u);
            }
        }
    };
    // two accumulator sets in turn: fragment f's MFMAs run while fragment f - 1's epilogue is issued
    f32x4 acc2[2][MB];
    mma(F0, acc2[0]);
    static_for<F1 - F0 - 1>([&](auto fc) __attribute__((always_inline)) {
        constexpr int i = decltype(fc)::value + 1;
        mma(F0 + i, acc2[i & 1]);
        epi(F0 + i - 1, acc2[(i - 1) & 1]);
    });
    epi(F1 - 1, acc2[(F1 - F0 - 1) & 1]);
}

// ---- eight waves: conv 24 -> 24 (+ bias, PReLU), stage 1..4 of a pipeline: ring stage-1 -> ring stage;
//      part 1's light back wave (TAIL): ring 4 -> u8 frame -----------------------------------------------------------------------------
//      TAIL waves take fragments [F0, F1) of the last layer; LOAD (part 1's front wave): also the rows of `mid` in -> ring 0 --------------
template <bool TAIL, int F0, int F1, bool LOAD>
__device__ __forceinline__ void body(const Sub5Args& a, const Lds L, const int wave, const int stage, const int li, const int lag,
                                     const int lane, const int xoff, const int nrows, const int nsteps)
{
    constexpr int KS = 7, MB = TAIL ? 1 : 2;
    const int p = lane & 15, o = lane >> 4, pix = pix_of(p);
    [[maybe_unused]] const bool stamp = a.dbg != nullptr && blockIdx.x == 0 && lane == 0;
    half8 wgt[KS][MB];
#pragma unroll
    for (int ks = 0; ks < KS; ++ks)
#pragma unroll
        for (int m = 0; m < MB; ++m) wgt[ks][m] = ((const half8*)a.wpk[li])[(ks * MB + m) * 64 + lane];
    // per k-step: the LDS address this lane's K octet is read from, for the row the wave works on next; one ring row on per step
    const unsigned in_ring = (unsigned)(stage - 1) * RINGB;
    unsigned adr[KS];
#pragma unroll
    for (int ks = 0; ks < KS; ++ks) {
        const int oct = min((int)(o == 0 ? SUB16_OCTET[ks][0] : o == 1 ? SUB16_OCTET[ks][1] : o == 2 ? SUB16_OCTET[ks][2] : SUB16_OCTET[ks][3]), 26);
        const int tap = oct / 3;
        adr[ks] = in_ring + ((tap / 3 - 1) & 3) * ROWB + (tap % 3 + pix) * PIXB + (oct % 3) * 16;      // row d = 0: tap row dy reads ring row (dy - 1) & 3
    }
    const float* const myprm = L.prm + li * 96;
    const Prm q = params(myprm, o);
    const f32x4 binit[2] = {*(const f32x4*)(myprm + 4 * o), f32x4{myprm[16 + 2 * o], myprm[17 + 2 * o], 0.f, 0.f}};
    char* const out_ring = L.pipe + stage * RINGB;

    // LOAD: row t of `mid` (all 66 ring columns; outside the plane: zeros) was requested a step ago and goes into ring 0 now; row
    // t + 1 is requested behind it -- HBM latency has a whole step to pass
    constexpr int LUNITS = ROWPX * 3, NK = LOAD ? (LUNITS + 63) / 64 : 1;
    uint4 lv[NK];
    [[maybe_unused]] auto fetch_mid = [&](int r) {
#pragma unroll
        for (int k = 0; k < NK; ++k) lv[k] = make_uint4(0, 0, 0, 0);
        if (r < nrows) {
            const int2 e = L.rows[r];
            const int y = e.x >> 1, x0c = e.y + xoff;
            if (y >= 0 && y < a.h) {
                const char* const src = a.mid + ((size_t)y * a.w + (x0c - 1)) * PIXB;
#pragma unroll
                for (int k = 0; k < NK; ++k) {
                    const int u = lane + 64 * k, X = x0c - 1 + u / 3;
                    if (u < LUNITS && X >= 0 && X < a.w) lv[k] = *(const uint4*)(src + (ptrdiff_t)u * 16);
                }
            }
        }
    };
    if constexpr (LOAD) fetch_mid(0);
    int2 desc = make_int2(0, 0);        // the descriptor of the next step's row, fetched a step ahead
    for (int t = 0; t < nsteps; ++t) {
        S5_STAMP(0);
        if constexpr (LOAD) {
            if (t < nrows) {
                char* const dst = L.pipe + (t & 3) * ROWB;
#pragma unroll
                for (int k = 0; k < NK; ++k) {
                    const int u = lane + 64 * k;
                    if (u < LUNITS) *(uint4*)(dst + u * 16) = lv[k];
                }
            }
            fetch_mid(t + 1);
        }
        const int d = t - lag;
        if (d >= 0 && d < nrows) {
            const int ye = __builtin_amdgcn_readfirstlane(desc.x), x0c = __builtin_amdgcn_readfirstlane(desc.y) + xoff;
            const int yy = ye >> 1;
            const bool row_in = yy >= 0 && yy < a.h;
            char* const px = out_ring + (d & 3) * ROWB + (pix + 1) * PIXB;
            uint8_t* const dst = a.dst + (size_t)yy * a.dst_stride + (size_t)(x0c + pix) * 3;
            unsigned r8[4] = {0, 0, 0, 0};
            if constexpr (TAIL) {
                // the residual: this row's input pixels straight from the u8 frame (lane group o = channel o: one byte each),
                // requested before the MFMAs and used behind them
                if ((ye & 1) && row_in && o < 3) {
                    const uint8_t* sp = a.src + (size_t)yy * a.src_stride + (size_t)(x0c + pix) * 3 + o;
#pragma unroll
                    for (int f = F0; f < F1; ++f) {
                        const int X = x0c + 16 * f + pix;
                        if (X >= 0 && X < a.w) r8[f] = (unsigned)sp[f * 48];
                    }
                }
            }
            if (!TAIL && row_in && x0c >= 0 && x0c + S5_WC <= a.w)
                rowf<TAIL, false, KS, MB, F0, F1>(L.pipe, px + 8 * o, px + 32 + 4 * o, dst, adr, wgt, binit, q, r8, x0c, a.w, row_in, (ye & 1) != 0, pix, o);
            else
                rowf<TAIL, true, KS, MB, F0, F1>(L.pipe, px + 8 * o, px + 32 + 4 * o, dst, adr, wgt, binit, q, r8, x0c, a.w, row_in, (ye & 1) != 0, pix, o);
            // next row: every address one ring row on, wrapping after the fourth (the increments are scalars: they depend on the
            // window row dy an octet comes from only)
            int inc[3];
#pragma unroll
            for (int dy = 0; dy < 3; ++dy) inc[dy] = ((d + dy - 1) & 3) == 3 ? -3 * ROWB : ROWB;
#pragma unroll
            for (int ks = 0; ks < KS; ++ks) {
                const int d0 = dy_of(ks, 0), d1 = dy_of(ks, 1), d2 = dy_of(ks, 2), d3 = dy_of(ks, 3);
                if (d0 == d1 && d1 == d2 && d2 == d3) adr[ks] += (unsigned)inc[d0];
                else adr[ks] += (unsigned)(o == 0 ? inc[d0] : o == 1 ? inc[d1] : o == 2 ? inc[d2] : inc[d3]);
            }
        }
        S5_STAMP(2);
        if (d + 1 >= 0 && d + 1 < nrows) desc = L.rows[d + 1];
        barrier();
    }
}

}  // namespace s5

template <int PART>
__global__ __launch_bounds__(64 * s5::NW, 1) UVA_NO_PK_F32 void sub5_kernel(Sub5Args a)
{
    using namespace s5;
    extern __shared__ __attribute__((aligned(16))) char smem[];
    const int wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
    const int lane = threadIdx.x & 63;
    const int nrows = __builtin_amdgcn_readfirstlane(a.nrows[blockIdx.x]);
    if (nrows <= 0) return;
    float* const prm = (float*)(smem + 2 * PIPEB);
    int2* const rows = (int2*)(smem + 2 * PIPEB + PRMB);
    // this workgroup's row descriptors live in LDS, 8 bytes each: {2y + emit, x0 of the first pipeline}
    {
        const uint4* const grows = a.rows + (size_t)blockIdx.x * a.max_rows;
        for (int i = threadIdx.x; i < nrows; i += 64 * NW) {
            const uint4 e = grows[i];
            rows[i] = make_int2((int)e.x * 2 + (int)(e.z & 1), (int)e.y);
        }
    }
    // rings start as zeros (margins and pipeline fill are never written: no NaN patterns may sit there)
    for (int i = threadIdx.x; i < 2 * PIPEB / 16; i += 64 * NW) ((uint4*)smem)[i] = make_uint4(0, 0, 0, 0);
    if (wave < S5_NL && lane < 32) {
        prm[wave * 96 + lane] = a.bias[wave][lane];
        prm[wave * 96 + 32 + lane] = (PART == 1 && wave == S5_NL - 1) ? 0.f : a.slope[wave][lane];
    }
    __syncthreads();
    // every wave runs the same number of steps = barriers, whatever code it sits in (part 0: the last row leaves 11 steps after it
    // went in, part 1: 10)
    const int nsteps = (nrows + 12 + 1) & ~1;
    // waves w, w+4, w+8 share a SIMD: two trunk layers and one light wave on each
    const int pipe = wave < 8 ? wave >> 2 : wave & 1;
    const Lds L{smem + pipe * PIPEB, prm, rows};
    const int xoff = pipe * S5_VALID;
    if (wave < 8) {
        const int stage = (wave & 3) + 1;
        if (PART == 0) body<false, 0, 4, false>(a, L, wave, stage, stage, 2 * stage + 2, lane, xoff, nrows, nsteps);   // layer `stage`, two rows behind the one above
        else body<false, 0, 4, false>(a, L, wave, stage, stage - 1, 2 * stage, lane, xoff, nrows, nsteps);              // layer 4 + stage (ring 0 is loaded, not computed)
    } else if (wave < 10) {
        if (PART == 0) head<0, 2, true>(a, L, wave, lane, xoff, nrows, nsteps);
        else body<true, 0, S5_TAIL_SPLIT, true>(a, L, wave, S5_NL, S5_NL - 1, 2 * S5_NL, lane, xoff, nrows, nsteps);    // conv 24 -> 3 reads ring 4
    } else {
        if (PART == 0) head<2, 4, false>(a, L, wave, lane, xoff, nrows, nsteps);
        else body<true, S5_TAIL_SPLIT, 4, false>(a, L, wave, S5_NL, S5_NL - 1, 2 * S5_NL, lane, xoff, nrows, nsteps);
    }
}

}  // namespace uva
