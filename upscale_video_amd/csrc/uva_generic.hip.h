// uva_generic.hip.h -- device executor for generic ncnn graphs (uva_generic.h): the `-m r` path of the
// reference (models/4x_Valar_v1.param, upscale/upscale_processing.py:913-916) and anything else made of
// the layer types listed there.  Functional first: one kernel per layer, every blob a zero-bordered fp16
// NHWC array of its own (channels padded to a multiple of 32, so every 3x3 / 1x1 convolution is the same
// implicit GEMM on v_mfma_f32_16x16x32_f16 with K = taps x padded input channels and no bounds checks),
// fp32 accumulate, bias / LeakyReLU in fp32.  Not fused, not LDS-tiled: the SRVGGNetCompact graphs -- the
// hot path -- never come here.
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>

#include <map>
#include <tuple>
#include <vector>

#include "uva_generic.h"
#include "uva_kernels.hip.h"
#include "uva_rdb.hip.h"

namespace uva {

// blob b of an h x w plane at scale s: [(h*s + 3)][(w*s + 2)][cpad] fp16; pixel (y, x) at row y+1, col x+1;
// one guard row behind the bottom border row (the convolution reads 16-pixel groups past the right edge)
struct GBuf {
    _Float16* p = nullptr;
    int h = 0, w = 0, c = 0, cpad = 0;
    size_t elems() const { return (size_t)(h + 3) * (w + 2) * cpad; }
    size_t pitch() const { return (size_t)(w + 2) * cpad; }
};

__device__ __forceinline__ size_t gb_off(int w, int cpad, int y, int x) { return ((size_t)(y + 1) * (w + 2) + (x + 1)) * cpad; }

// from_pixels(PIXEL_BGR) + substract_mean_normalize([], [1/255]*3) (upscale_processing.py:265-273, :437-445)
__global__ void g_input_u8(const uint8_t* src, size_t stride, int y0, int x0, int h, int w, _Float16* out, int cpad)
{
    const int x = blockIdx.x * blockDim.x + threadIdx.x, y = blockIdx.y;
    if (x >= w) return;
    const uint8_t* s = src + (size_t)(y0 + y) * stride + (size_t)(x0 + x) * 3;
    _Float16* o = out + gb_off(w, cpad, y, x);
    const float norm = (float)(1 / 255.0);
    for (int c = 0; c < 3; ++c) o[c] = (_Float16)((float)s[c] * norm);
}

__global__ void g_input_f32(const float* src, int h, int w, _Float16* out, int cpad)
{
    const int x = blockIdx.x * blockDim.x + threadIdx.x, y = blockIdx.y;
    if (x >= w) return;
    _Float16* o = out + gb_off(w, cpad, y, x);
    for (int c = 0; c < 3; ++c) o[c] = (_Float16)src[((size_t)c * h + y) * w + x];
}

// ncnn convolution.cpp (stride 1, 'same' zero padding, optional bias, optional fused LeakyReLU):
// one wave = 64 pixels of a row (four 16-pixel B fragments) x up to 64 output channels (four A fragments), so
// every operand fetched from L2 feeds four MFMAs; a workgroup = 4 consecutive rows.
template <int KSIZE>
__global__ __launch_bounds__(256) void g_conv(const _Float16* in, int cin_pad, const half8* wpk, const float* bias, _Float16* out,
                                              int cout, int cout_pad, int out_cpad, int h, int w, int has_act, float slope)
{
    const int wave = threadIdx.x >> 6, lane = threadIdx.x & 63;
    const int x0 = blockIdx.x * 64, y = blockIdx.y * 4 + wave;
    if (y >= h) return;
    const int mbn = cout_pad / 16, mb0 = blockIdx.z * 4, nmb = min(4, mbn - mb0);
    const int c32n = cin_pad / 32, p = lane & 15, oct = lane >> 4;
    const int nfr = min(4, (w - x0 + 15) / 16);
    f32x4 acc[4][4];
#pragma unroll
    for (int f = 0; f < 4; ++f)
#pragma unroll
        for (int m = 0; m < 4; ++m) acc[f][m] = f32x4{0.f, 0.f, 0.f, 0.f};
    constexpr int TAPS = KSIZE * KSIZE;
    for (int tap = 0; tap < TAPS; ++tap) {
        const int dy = KSIZE == 3 ? tap / 3 - 1 : 0, dx = KSIZE == 3 ? tap % 3 - 1 : 0;
        const _Float16* src = in + gb_off(w, cin_pad, y + dy, x0 + p + dx) + 8 * oct;
        for (int c32 = 0; c32 < c32n; ++c32) {
            const half8* wp = wpk + ((size_t)(tap * c32n + c32) * mbn + mb0) * 64 + lane;
            half8 a[4], b[4];
#pragma unroll
            for (int m = 0; m < 4; ++m) a[m] = wp[(m < nmb ? m : 0) * 64];
#pragma unroll
            for (int f = 0; f < 4; ++f) b[f] = *(const half8*)(src + (size_t)(f < nfr ? f : 0) * 16 * cin_pad + 32 * c32);
#pragma unroll
            for (int f = 0; f < 4; ++f)
#pragma unroll
                for (int m = 0; m < 4; ++m) acc[f][m] = __builtin_amdgcn_mfma_f32_16x16x32_f16(a[m], b[f], acc[f][m], 0, 0, 0);
        }
    }
#pragma unroll
    for (int f = 0; f < 4; ++f) {
        const int x = x0 + 16 * f + p;
        if (f >= nfr || x >= w) continue;
        _Float16* o = out + gb_off(w, out_cpad, y, x);
#pragma unroll
        for (int m = 0; m < 4; ++m) {
            if (m >= nmb) continue;
            const int ch = 16 * (mb0 + m) + 4 * oct;
#pragma unroll
            for (int j = 0; j < 4; ++j) {
                if (ch + j >= cout) continue;
                float v = acc[f][m][j] + bias[ch + j];
                if (has_act) v = v > 0.f ? v : v * slope;      // ncnn activation_type 2: LeakyReLU
                o[ch + j] = (_Float16)v;
            }
        }
    }
}

// ---- g_conv3_lds: the 3x3 convolution of the generic executor with both operands through LDS ---------------------
// g_conv<3> above fetches every B fragment nine times from L2 (once per tap) and every weight fragment once per wave:
// it runs at ~6 % of the MFMA peak.  Here a workgroup (4 waves) owns an 8 x 32 output tile:
//   * its 10 x 34 halo tile of the input array goes to LDS once (all cin channels; pixel stride cin*2 + 16 bytes: an odd
//     number of 16-byte units, so that the 16 pixels of a fragment land on 16 different units);
//   * the weights go through LDS one tap at a time (cin/32 k-steps x 1 KiB per 16 output channels), the next tap's
//     already on their way from L2 in registers while this tap's k-steps run: two workgroup barriers per tap;
//   * a wave computes 2 rows x 32 pixels x all output channels (4 fragments x MBN blocks of v_mfma_f32_16x16x32_f16).
// LDS reads are conflict-free for the same reason as in sub10_kernel: MFMA column p holds pixel gpix(p) (even pixels in
// the lanes {0-3,12-15}, odd ones in {4-11}: `ds_read_b128` is served in groups of 8 lanes of K-octet group o and 8 of
// o^1), and the octet groups o = 0..3 of a k-step read the 16-byte units {0, 2, 1, 3} of the 64 bytes of their 32
// channels -- o and o^1 two units apart (pack_generic(..., lds_order = true) arranges the weights to match).
// The output goes to channels [out_coff, out_coff + cout) of an array whose pixel stride may be larger (a dense block's
// concatenation buffer: the Concat layers of an RRDB then cost nothing).
struct GConvArgs {
    const _Float16* in;           // zero-bordered array, pixel stride in_stride elements
    int in_stride, cin_pad;       // channels read: [0, cin_pad), a multiple of 32
    const half8* wpk;             // pack_generic(lds_order) image: [tap][cin_pad/32][cout_pad/16][64][8]
    const float* bias;            // [cout_pad] (zero padded) or nullptr
    _Float16* out;
    int out_stride, out_coff, cout;
    int h, w;
    int has_act;
    float slope;
    // the element-wise sum that follows the convolution, done while its result leaves (the sum layer is then skipped):
    // out = x * ca + y * cb with (x, y) = (result, res) or, res_first, (res, result) -- g_axpby's expression, its operand
    // order and its rounding of the convolution's result to fp16 first: bit for bit what the two launches give
    const _Float16* res;          // nullptr: no sum
    int res_stride, res_first;
    float ca, cb;
    // the graph's LAST convolution on the u8 route (cout not a multiple of 8: 3 channels): g_output_u8's arithmetic on the
    // value rounded to fp16 -- rint(v * 255), clamped -- straight into the frame, the tile's core [cy0, cy1) x [cx0, cx1)
    // only, at (dy0 + y, dx0 + x); the fp16 array is then not written at all
    uint8_t* u8dst;               // nullptr: the fp16 array as usual
    size_t u8stride;
    int dy0, dx0, cy0, cy1, cx0, cx1;
};
constexpr int GC_TH = 8, GC_TW = 32;
constexpr int GC_CH = 1;          // input channels go through LDS 32 at a time (one k-step): 33 KB and 80-100 registers per
                                  // workgroup, four of them share a CU (64 at a time: two or three, -14 % on Valar)
// (ksize 1: the same kernel without the halo and with a single tap -- the RRDBs' 1x1 residual convolutions)
inline size_t g_conv3_lds_bytes(int cin_pad, int mbn, int ksize = 3, bool wg = false, int nw = 4)
{
    const int th = 2 * nw;                                                                             // tile rows: two per wave
    const size_t npix = (size_t)(th + ksize - 1) * (GC_TW + ksize - 1);
    const int cn = std::min(GC_CH, cin_pad / 32);
    const size_t work = npix * (cn * 64 + 16) + (wg ? 0 : (size_t)ksize * cn * mbn * 1024);            // halo tile chunk + one row of taps
    const size_t stage = (size_t)th * GC_TW * (mbn * 32 + 16);                                         // the epilogue's output staging tile
    return work > stage ? work : stage;
}
__device__ __forceinline__ int gpix(int p) { return p < 4 ? 2 * p : p >= 12 ? 2 * (p - 8) : 2 * (p - 4) + 1; }

// WG: the weights do not go through LDS: every wave fetches its fragments of the next row of taps from L2 / L1 into
// registers while the current row's MFMAs run.  No weight staging, no barrier per row of taps (two per channel chunk
// remain, around the tile), 6 KB less LDS.
// NW waves own a 2 NW x 32 output tile (8 x 32, or 16 x 32: its halo is 1.20 instead of 1.33 times the tile -- the
// convolutions of a dense block are bound by re-reading their inputs).
template <int MBN, int KSZ = 3, bool WG = false, int NW = 4>
__global__ __launch_bounds__(64 * NW) void g_conv3_lds(GConvArgs a)
{
    static_assert(GC_CH == 1, "32-channel chunks: four 16-byte units per pixel");
    static_assert(!WG || KSZ == 3, "weights from global memory: 3x3");
    constexpr int NT = 64 * NW, TH = 2 * NW;
    constexpr int GC_PH = TH + KSZ - 1, GC_PW = GC_TW + KSZ - 1, GC_NPIX = GC_PH * GC_PW, ORG = KSZ == 3 ? 0 : 1;
    constexpr int RB = GC_PH / 2;                                // tile rows per load batch
    static_assert(GC_PH % 2 == 0, "two load batches");
    constexpr int ROWU = GC_PW * 4, HU = RB * ROWU, KT = (HU + NT - 1) / NT;   // 16-byte units per tile row, per batch, per thread
    extern __shared__ __attribute__((aligned(16))) char gsm[];
    const int tid = threadIdx.x, wave = tid >> 6, lane = tid & 63;
    const int p = lane & 15, o = lane >> 4, pix = gpix(p);
    const int x0 = blockIdx.x * GC_TW, y0 = blockIdx.y * TH;
    const int c32n = a.cin_pad / 32;
    const int cn_max = min(GC_CH, c32n);
    char* const tile = gsm;
    char* const wbuf = gsm + (size_t)GC_NPIX * (cn_max * 64 + 16);   // one ROW of taps of the current channel chunk

    // The input channels go through LDS in chunks of 32 * GC_CH (the tile of a 192-channel convolution would fill the LDS:
    // one workgroup per CU, its load, compute and store phases in series -- with 33 KiB per workgroup four share a
    // CU and overlap them), the weights one row of taps (KSZ taps x chunk) at a time, the next row's on their way in
    // registers while this one is used.  Stage s = (chunk, tap row).
    constexpr int WMAX = (KSZ * GC_CH * MBN * 64 + NT - 1) / NT;   // 16-byte units per thread and weight stage
    const int nstage = ((c32n + GC_CH - 1) / GC_CH) * KSZ;
    half8 wreg[WMAX];
    auto wfetch = [&](int s) {
        const int c0 = (s / KSZ) * GC_CH, cn = min(GC_CH, c32n - c0), tr = s % KSZ;
        const int per_tap = cn * MBN * 64;                       // units per tap of this chunk
#pragma unroll
        for (int k = 0; k < WMAX; ++k) {
            const int j = tid + NT * k;
            if (j < KSZ * per_tap) {
                const int t = j / per_tap, r = j - t * per_tap;
                wreg[k] = a.wpk[((size_t)(tr * KSZ + t) * c32n + c0) * MBN * 64 + r];
            }
        }
    };
    auto wstore = [&](int s) {
        const int c0 = (s / KSZ) * GC_CH, cn = min(GC_CH, c32n - c0);
        const int per_tap = cn * MBN * 64;
#pragma unroll
        for (int k = 0; k < WMAX; ++k) {
            const int j = tid + NT * k;
            if (j < KSZ * per_tap) *(half8*)(wbuf + (size_t)j * 16) = wreg[k];
        }
    };

    f32x4 acc[4][MBN];
#pragma unroll
    for (int f = 0; f < 4; ++f)
#pragma unroll
        for (int m = 0; m < MBN; ++m) acc[f][m] = f32x4{0.f, 0.f, 0.f, 0.f};
    const int unit_of_o = o == 0 ? 0 : o == 1 ? 2 : o == 2 ? 1 : 3;

    // half a tile chunk in registers: unit j of the batch is row j / ROWU, pixel (j % ROWU) / 4, 16-byte unit j % 4
    uint4 tv2[2][KT];
    auto tfetch = [&](int c0, int half) {
        uint4 (&tv)[KT] = tv2[half];
#pragma unroll
        for (int k = 0; k < KT; ++k) {
            const int j = tid + NT * k, rr = j / ROWU, q = j - rr * ROWU, c = q >> 2, u = q & 3;
            const int ay = y0 + ORG + RB * half + rr;
            tv[k] = make_uint4(0, 0, 0, 0);
            if (j < HU && ay <= a.h + 1 && x0 + ORG + c <= a.w + 1)
                tv[k] = *(const uint4*)(a.in + ((size_t)ay * (a.w + 2) + x0 + ORG + c) * a.in_stride + 32 * c0 + 8 * u);
        }
    };
    auto tstore = [&](int half, int pstride) {
        uint4 (&tv)[KT] = tv2[half];
#pragma unroll
        for (int k = 0; k < KT; ++k) {
            const int j = tid + NT * k, rr = j / ROWU, q = j - rr * ROWU, c = q >> 2, u = q & 3;
            if (j < HU) *(uint4*)(tile + (size_t)((RB * half + rr) * GC_PW + c) * pstride + 16 * u) = tv[k];
        }
    };

    half8 wq[KSZ][MBN];                                          // WG: this row of taps' weight fragments
    auto wload = [&](int st, int t, half8 (&wv)[MBN]) {
        const int c0 = st / KSZ, tr = st % KSZ;
#pragma unroll
        for (int m = 0; m < MBN; ++m) wv[m] = a.wpk[((size_t)(tr * KSZ + t) * c32n + c0) * MBN * 64 + m * 64 + lane];
    };
    if constexpr (WG) {
#pragma unroll
        for (int t = 0; t < KSZ; ++t) wload(0, t, wq[t]);
    } else {
        wfetch(0);
    }
    for (int s = 0; s < nstage; ++s) {
        const int c0 = (s / KSZ) * GC_CH, cn = min(GC_CH, c32n - c0), tr = s % KSZ;
        const int pstride = cn * 64 + 16;                        // bytes per pixel of this chunk's tile: an odd number of units
        if (tr == 0) {
            // halo tile of this chunk: array rows y0 .. y0+9, columns x0 .. x0+33 (the array carries a one-pixel zero
            // border: pixel (y, x) sits at row y+1, column x+1), channels 32*c0 .. +32*cn; outside the array: zeros.  A
            // tile row is GC_PW consecutive pixels of the array.  The first chunk's rows are fetched here, half of them in flight at a time; the
            // first half of every later chunk's rows has been on its way since the previous chunk's tile was stored
            // (tfetch below), so that only the second half's latency is left in the open.
            if (s > 0) __syncthreads();                          // the previous chunk's tile has been used up
            if (s == 0) { tfetch(c0, 0); tfetch(c0, 1); }
            tstore(0, pstride);
            tstore(1, pstride);
            if (s + KSZ < nstage) { tfetch(c0 + GC_CH, 0); tfetch(c0 + GC_CH, 1); }
            if constexpr (WG) __syncthreads();                   // the tile is in place
        } else if constexpr (!WG) {
            __syncthreads();                                     // everybody is done with the previous row of taps
        }
        if constexpr (!WG) {
            wstore(s);
            __syncthreads();
            if (s + 1 < nstage) wfetch(s + 1);
        }

        // fragment f = 2*n + c: row 2*wave + n, columns 16*c .. 16*c + 15 of the tile's interior
        unsigned fbase[4];
#pragma unroll
        for (int f = 0; f < 4; ++f)
            fbase[f] = (unsigned)(((2 * wave + (f >> 1) + tr) * GC_PW + 16 * (f & 1) + pix) * pstride + unit_of_o * 16);
        if constexpr (WG) {
            // one k-step per tap column; a slot is refilled with the same column of the NEXT row of taps right after use
#pragma unroll
            for (int t = 0; t < KSZ; ++t) {
                half8 bv[4];
#pragma unroll
                for (int f = 0; f < 4; ++f) bv[f] = *(const half8*)(tile + fbase[f] + t * pstride);
#pragma unroll
                for (int f = 0; f < 4; ++f)
#pragma unroll
                    for (int m = 0; m < MBN; ++m) acc[f][m] = __builtin_amdgcn_mfma_f32_16x16x32_f16(wq[t][m], bv[f], acc[f][m], 0, 0, 0);
                if (s + 1 < nstage) wload(s + 1, t, wq[t]);
            }
            continue;
        }
        // k-steps of this stage: (tap column t, 32-channel group c); operands of the next one are read while this one's
        // MFMAs run
        const int nk = KSZ * cn;
        half8 wa[2][MBN], bf[2][4];
        auto rd = [&](int k, half8 (&wv)[MBN], half8 (&bv)[4]) {
            const int t = k / cn, c = k - t * cn;
            const char* const wcur = wbuf + (size_t)(t * cn + c) * MBN * 1024;
#pragma unroll
            for (int m = 0; m < MBN; ++m) wv[m] = *(const half8*)(wcur + m * 1024 + lane * 16);
#pragma unroll
            for (int f = 0; f < 4; ++f) bv[f] = *(const half8*)(tile + fbase[f] + t * pstride + c * 64);
        };
        auto mm = [&](const half8 (&wv)[MBN], const half8 (&bv)[4]) {
#pragma unroll
            for (int f = 0; f < 4; ++f)
#pragma unroll
                for (int m = 0; m < MBN; ++m) acc[f][m] = __builtin_amdgcn_mfma_f32_16x16x32_f16(wv[m], bv[f], acc[f][m], 0, 0, 0);
        };
        rd(0, wa[0], bf[0]);
        int k = 0;
        for (; k + 2 <= nk; k += 2) {
            rd(k + 1, wa[1], bf[1]);
            mm(wa[0], bf[0]);
            if (k + 2 < nk) rd(k + 2, wa[0], bf[0]);
            mm(wa[1], bf[1]);
        }
        if (k < nk) mm(wa[0], bf[0]);
    }
    // bias, LeakyReLU (ncnn activation_type 2), fp16; lane (o, p): channels 16m + 4o .. +3 of pixel gpix(p).  Through LDS
    // (the input tile is dead by now): every pixel's channels then leave as consecutive 16-byte units -- a whole tile row
    // in one piece when the output array is dense -- instead of 8-byte pieces at a pixel stride.
    __syncthreads();
    constexpr int OUTB = MBN * 32 + 16;                           // bytes per pixel in the staging tile (odd number of units)
    char* const stage = gsm;
#pragma unroll
    for (int f = 0; f < 4; ++f) {
        const int px = (2 * wave + (f >> 1)) * GC_TW + 16 * (f & 1) + pix;
#pragma unroll
        for (int m = 0; m < MBN; ++m) {
            const int ch = 16 * m + 4 * o;
            float v[4];
#pragma unroll
            for (int j = 0; j < 4; ++j) {
                v[j] = acc[f][m][j] + (a.bias ? a.bias[min(ch + j, 16 * MBN - 1)] : 0.f);
                if (a.has_act) v[j] = v[j] > 0.f ? v[j] : v[j] * a.slope;
            }
            *(half4*)(stage + px * OUTB + ch * 2) = half4{(_Float16)v[0], (_Float16)v[1], (_Float16)v[2], (_Float16)v[3]};
        }
    }
    __syncthreads();
    if (a.cout % 8 == 0) {
        const int upp = a.cout / 8;                               // 16-byte units per pixel
        for (int i = tid; i < TH * GC_TW * upp; i += NT) {
            const int px = i / upp, u = i - px * upp;
            const int y = y0 + px / GC_TW, x = x0 + px % GC_TW;
            if (y < a.h && x < a.w) {
                const size_t pos = (size_t)(y + 1) * (a.w + 2) + (x + 1);
                uint4 v = *(const uint4*)(stage + px * OUTB + 16 * u);
                if (a.res) {
                    const half8 c = __builtin_bit_cast(half8, v), r = *(const half8*)(a.res + pos * a.res_stride + 8 * u);
                    const half8 xx = a.res_first ? r : c, yy = a.res_first ? c : r;
                    half8 o8;
#pragma unroll
                    for (int e = 0; e < 8; ++e) o8[e] = g_axpby1((float)xx[e], a.ca, (float)yy[e], a.cb);
                    v = __builtin_bit_cast(uint4, o8);
                }
                *(uint4*)(a.out + pos * a.out_stride + a.out_coff + 8 * u) = v;
            }
        }
    } else {
        for (int i = tid; i < TH * GC_TW * a.cout; i += NT) {
            const int px = i / a.cout, c = i - px * a.cout;
            const int y = y0 + px / GC_TW, x = x0 + px % GC_TW;
            if (a.u8dst) {
                if (y >= a.cy0 && y < a.cy1 && x >= a.cx0 && x < a.cx1) {
                    float q = __builtin_rintf((float)*(const _Float16*)(stage + px * OUTB + 2 * c) * 255.0f);
                    q = fminf(fmaxf(q, 0.f), 255.f);
                    a.u8dst[(size_t)(a.dy0 + y) * a.u8stride + (size_t)(a.dx0 + x) * a.cout + c] = (uint8_t)q;
                }
            } else if (y < a.h && x < a.w)
                a.out[((size_t)(y + 1) * (a.w + 2) + (x + 1)) * a.out_stride + a.out_coff + c] = *(const _Float16*)(stage + px * OUTB + 2 * c);
        }
    }
}

// BinaryOp ADD (ca = cb = 1) and Eltwise SUM with coefficients: whole arrays (0*ca + 0*cb keeps the border zero)
__global__ void g_axpby(const half8* a, float ca, const half8* b, float cb, half8* out, size_t n8)
{
    const size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n8) return;
    const half8 x = a[i], y = b[i];
    half8 r;
#pragma unroll
    for (int e = 0; e < 8; ++e) r[e] = g_axpby1((float)x[e], ca, (float)y[e], cb);
    out[i] = r;
}

// the same on channel ranges of arrays with different pixel strides (operands or result inside a dense chain's shared
// array): c8 octets of channels per pixel, every pixel of the bordered array
__global__ void g_axpby_strided(const _Float16* a, int sa, float ca, const _Float16* b, int sb, float cb, _Float16* out, int so, int c8,
                                size_t npix)
{
    const size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= npix * c8) return;
    const size_t pix = i / c8;
    const int k = (int)(i - pix * c8);
    const half8 x = *(const half8*)(a + pix * sa + 8 * k), y = *(const half8*)(b + pix * sb + 8 * k);
    half8 r;
#pragma unroll
    for (int e = 0; e < 8; ++e) r[e] = g_axpby1((float)x[e], ca, (float)y[e], cb);
    *(half8*)(out + pix * so + 8 * k) = r;
}

// Concat along channels: one input's c_in channels (a multiple of 8) into [c_off, c_off + c_in) of the output
__global__ void g_concat_part(const _Float16* in, int cpad_in, int c_in, _Float16* out, int cpad_out, int c_off, size_t npix)
{
    const int oct_n = c_in / 8;
    const size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= npix * oct_n) return;
    const size_t pix = i / oct_n;
    const int o = (int)(i - pix * oct_n);
    *(half8*)(out + pix * cpad_out + c_off + 8 * o) = *(const half8*)(in + pix * cpad_in + 8 * o);
}

// ncnn interp.cpp resize_type 1 (nearest) with an integer factor: out(Y, X) = in(Y / f, X / f)
__global__ void g_interp_nearest(const _Float16* in, int hi, int wi, int cpad, _Float16* out, int f)
{
    const int ho = hi * f, wo = wi * f;
    const int oct_n = cpad / 8;
    const size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= (size_t)ho * wo * oct_n) return;
    const int o = (int)(i % oct_n);
    const size_t pix = i / oct_n;
    const int X = (int)(pix % wo), Y = (int)(pix / wo);
    *(half8*)(out + gb_off(wo, cpad, Y, X) + 8 * o) = *(const half8*)(in + gb_off(wi, cpad, Y / f, X / f) + 8 * o);
}

// ncnn prelu.cpp: x < 0 ? x * slope[c] : x
__global__ void g_prelu(const _Float16* in, const float* slopes, _Float16* out, int c, int cpad, size_t npix)
{
    const size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= npix * c) return;
    const size_t pix = i / c;
    const int ch = (int)(i - pix * c);
    const float x = (float)in[pix * cpad + ch];
    out[pix * cpad + ch] = (_Float16)(x < 0.f ? x * slopes[ch] : x);
}

// ncnn pixelshuffle.cpp mode 0: out(c, Y, X) = in(c*f*f + (Y%f)*f + X%f, Y/f, X/f)
__global__ void g_pixelshuffle(const _Float16* in, int hi, int wi, int cpad_in, _Float16* out, int c_out, int cpad_out, int f)
{
    const int ho = hi * f, wo = wi * f;
    const size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= (size_t)ho * wo * c_out) return;
    const int c = (int)(i % c_out);
    const size_t pix = i / c_out;
    const int X = (int)(pix % wo), Y = (int)(pix / wo);
    out[gb_off(wo, cpad_out, Y, X) + c] = in[gb_off(wi, cpad_in, Y / f, X / f) + c * f * f + (Y % f) * f + (X % f)];
}

// np.array(mat_out).transpose(1, 2, 0) * 255 -> cv2 convertTo(CV_8U) (round half to even, saturate), the
// plane's core region only (process_tile's crop, upscale_processing.py:462-477)
__global__ void g_output_u8(const _Float16* in, int ho, int wo, int cpad, uint8_t* dst, size_t stride, int dy0, int dx0,
                            int cy0, int cy1, int cx0, int cx1)
{
    const int X = blockIdx.x * blockDim.x + threadIdx.x + cx0, Y = blockIdx.y + cy0;
    if (X >= cx1 || Y >= cy1) return;
    const _Float16* s = in + gb_off(wo, cpad, Y, X);
    uint8_t* d = dst + (size_t)(dy0 + Y) * stride + (size_t)(dx0 + X) * 3;
    for (int c = 0; c < 3; ++c) {
        float q = __builtin_rintf((float)s[c] * 255.0f);
        q = fminf(fmaxf(q, 0.f), 255.f);
        d[c] = (uint8_t)q;
    }
}

__global__ void g_output_f32(const _Float16* in, int ho, int wo, int cpad, float* dst)
{
    const int X = blockIdx.x * blockDim.x + threadIdx.x, Y = blockIdx.y;
    if (X >= wo) return;
    const _Float16* s = in + gb_off(wo, cpad, Y, X);
    for (int c = 0; c < 3; ++c) dst[((size_t)c * ho + Y) * wo + X] = (float)s[c];
}

// g_conv3_sw's work list: every plane (dims[2k], dims[2k+1] = its height, width) is cut into strips of `cols` columns, a
// strip into blocks of SW_R rows; the blocks -- plane after plane, strip after strip, top to bottom -- are dealt out to
// `grid` workgroups in contiguous runs of equal length (+-1); a run that crosses a strip's end becomes two segments.
// seg_begin has grid + 1 entries.
inline void sw_segments(const std::vector<int>& dims, int cols, int grid, std::vector<GSwSeg>& segs, std::vector<int>& seg_begin)
{
    struct Strip { int plane, c0, nb, h; };
    std::vector<Strip> strips;
    long long total = 0;
    for (size_t pl = 0; 2 * pl + 1 < dims.size(); ++pl) {
        const int h = dims[2 * pl], w = dims[2 * pl + 1], nb = (h + SW_R - 1) / SW_R;
        for (int c0 = 0; c0 < w; c0 += cols) { strips.push_back(Strip{(int)pl, c0, nb, h}); total += nb; }
    }
    segs.clear();
    seg_begin.assign(1, 0);
    // Several planes: like rdb_segments, a segment's start (two barriers and the wait for its first six rows) is counted as
    // SW_FILL blocks and the budget of blocks + starts is what the workgroups share evenly (UVA_RDB_DEAL=0: blocks alone).
    constexpr int SW_FILL = 1;
    static const bool by_steps = [] { const char* e = uva::debug_env("UVA_RDB_DEAL"); return !e || std::atoi(e) != 0; }();
    if (by_steps && dims.size() > 2) {
        auto deal = [&](long long T, bool emit) -> int {
            size_t si = 0;
            int y = 0, g = 0;
            if (emit) { segs.clear(); seg_begin.assign(1, 0); }
            while (si < strips.size()) {
                long long cap = T;
                while (si < strips.size() && cap > SW_FILL) {
                    const Strip& st = strips[si];
                    const int take = (int)std::min<long long>(st.nb - y, cap - SW_FILL);
                    if (emit) segs.push_back(GSwSeg{st.c0, y * SW_R, std::min(st.h, (y + take) * SW_R), st.plane});
                    cap -= take + SW_FILL;
                    y += take;
                    if (y == st.nb) { ++si; y = 0; }
                }
                ++g;
                if (emit) seg_begin.push_back((int)segs.size());
            }
            return g;
        };
        long long lo = total / grid + SW_FILL + 1, hi = total + SW_FILL * (long long)strips.size() + 1;
        while (lo < hi) {
            const long long mid = (lo + hi) / 2;
            if (deal(mid, false) <= grid) hi = mid; else lo = mid + 1;
        }
        deal(lo, true);
        while ((int)seg_begin.size() < grid + 1) seg_begin.push_back((int)segs.size());
        return;
    }
    size_t si = 0;
    long long sbase = 0;                       // blocks before strip si
    for (int g = 0; g < grid; ++g) {
        long long u = total * g / grid;
        const long long u1 = total * (g + 1) / grid;
        while (u < u1) {
            while (u >= sbase + strips[si].nb) { sbase += strips[si].nb; ++si; }
            const Strip& st = strips[si];
            const int b0 = (int)(u - sbase), b1 = (int)std::min<long long>(st.nb, b0 + (u1 - u));
            segs.push_back(GSwSeg{st.c0, b0 * SW_R, std::min(st.h, b1 * SW_R), st.plane});
            u += b1 - b0;
        }
        seg_begin.push_back((int)segs.size());
    }
}
inline void sw_segments(int h, int w, int cols, int grid, std::vector<GSwSeg>& segs, std::vector<int>& seg_begin)
{
    sw_segments(std::vector<int>{h, w}, cols, grid, segs, seg_begin);
}

// rdb4_kernel's work list: strips of 48 computed columns that own 42 of them (45 at the plane's left edge, everything up
// to the right edge in the last one), cut into row segments.
inline void rdb_strips(int h, int w, int plane, std::vector<RdbSeg>& strips)
{
    for (int own0 = 0; own0 < w;) {
        RdbSeg s{};
        s.c0 = own0 == 0 ? 0 : own0 - 3;
        s.own0 = own0;
        s.own1 = s.c0 + RA_C >= w ? w : s.c0 + RA_C - 3;
        s.yb = 0;
        s.ye = h;
        s.plane = plane;
        strips.push_back(s);
        own0 = s.own1;
    }
}
// Several planes in one launch (dims[2k], dims[2k+1] = height, width of plane k): the rows of all strips, plane after
// plane and strip after strip, are dealt out to the workgroups in contiguous runs of equal length; a run that crosses a
// strip's end becomes two segments (each costs RA_LAG steps of pipeline fill: with a whole frame's rows to share, runs are
// a couple of hundred rows long).  One plane: rdb_segments(h, w, ...) below, one segment per workgroup.
inline void rdb_segments(const std::vector<int>& dims, int grid, std::vector<RdbSeg>& segs, std::vector<int>& seg_begin);
inline void rdb_segments(int h, int w, int grid, std::vector<RdbSeg>& segs, std::vector<int>& seg_begin)
{
    std::vector<RdbSeg> strips;
    rdb_strips(h, w, 0, strips);
    // Every segment costs RA_LAG steps of pipeline fill, so a workgroup gets ONE: each strip is cut into k = grid / strips
    // equal row ranges (a few workgroups stay idle; with more strips than workgroups, whole strips are dealt out in turn).
    segs.clear();
    seg_begin.assign(1, 0);
    const int ns = (int)strips.size();
    if (ns >= grid) {
        for (int g = 0; g < grid; ++g) {
            for (int s = g; s < ns; s += grid) {
                RdbSeg sg = strips[s];
                sg.yb = 0;
                sg.ye = h;
                segs.push_back(sg);
            }
            seg_begin.push_back((int)segs.size());
        }
        return;
    }
    const int k = std::max(1, std::min(grid / ns, (h + 7) / 8));     // (ranges of fewer than ~8 rows are all pipeline fill)
    for (int g = 0; g < grid; ++g) {
        const int s = g / k, j = g - s * k;
        if (s < ns) {
            RdbSeg sg = strips[s];
            sg.yb = (int)((long long)h * j / k);
            sg.ye = (int)((long long)h * (j + 1) / k);
            if (sg.ye > sg.yb) segs.push_back(sg);
        }
        seg_begin.push_back((int)segs.size());
    }
}

inline void rdb_segments(const std::vector<int>& dims, int grid, std::vector<RdbSeg>& segs, std::vector<int>& seg_begin)
{
    if (dims.size() == 2) { rdb_segments(dims[0], dims[1], grid, segs, seg_begin); return; }
    std::vector<RdbSeg> strips;
    long long total = 0;
    for (size_t pl = 0; 2 * pl + 1 < dims.size(); ++pl) {
        const size_t n0 = strips.size();
        rdb_strips(dims[2 * pl], dims[2 * pl + 1], (int)pl, strips);
        total += (long long)(strips.size() - n0) * dims[2 * pl];
    }
    // What is dealt out evenly is STEPS, not rows: a segment costs its rows plus RDB_FILL steps of pipeline fill, and a
    // workgroup whose run crosses a strip's end has two segments (dealing rows alone left those 9 steps -- 4 % -- longer than
    // the others: the launch's tail).  A workgroup takes strip rows, in order, until its budget T is used up; T is the
    // smallest budget with which `grid` workgroups are enough.  UVA_RDB_DEAL=0: rows dealt evenly, the A/B switch.
    constexpr int RDB_FILL = 9;                // rdb4_kernel's RA_LAG
    static const bool by_steps = [] { const char* e = uva::debug_env("UVA_RDB_DEAL"); return !e || std::atoi(e) != 0; }();
    auto deal = [&](long long T, bool emit) -> int {
        size_t si = 0;
        int y = 0, g = 0;
        if (emit) { segs.clear(); seg_begin.assign(1, 0); }
        while (si < strips.size()) {
            long long cap = T;
            while (si < strips.size() && cap > RDB_FILL) {
                const int take = (int)std::min<long long>(strips[si].ye - y, cap - RDB_FILL);
                if (emit) { RdbSeg sg = strips[si]; sg.yb = y; sg.ye = y + take; segs.push_back(sg); }
                cap -= take + RDB_FILL;
                y += take;
                if (y == strips[si].ye) { ++si; y = 0; }
            }
            ++g;
            if (emit) seg_begin.push_back((int)segs.size());
        }
        return g;
    };
    if (by_steps) {
        long long lo = total / grid + RDB_FILL + 1, hi = total + RDB_FILL * (long long)strips.size() + 1;     // (a budget <= RDB_FILL buys no row)
        while (lo < hi) {
            const long long mid = (lo + hi) / 2;
            if (deal(mid, false) <= grid) hi = mid; else lo = mid + 1;
        }
        deal(lo, true);
        while ((int)seg_begin.size() < grid + 1) seg_begin.push_back((int)segs.size());
        return;
    }
    segs.clear();
    seg_begin.assign(1, 0);
    size_t si = 0;
    long long sbase = 0;                       // rows before strip si
    for (int g = 0; g < grid; ++g) {
        long long u = total * g / grid;
        const long long u1 = total * (g + 1) / grid;
        while (u < u1) {
            while (u >= sbase + strips[si].ye) { sbase += strips[si].ye; ++si; }
            RdbSeg sg = strips[si];
            const int y0 = (int)(u - sbase), y1 = (int)std::min<long long>(sg.ye, y0 + (u1 - u));
            sg.yb = y0;
            sg.ye = y1;
            segs.push_back(sg);
            u += y1 - y0;
        }
        seg_begin.push_back((int)segs.size());
    }
}

// ---------------------------------------------------------------------------------------------------
struct GenericDevice {
    struct ConvDev {
        half8* wpk = nullptr;
        half8* wpk_lds = nullptr;     // g_conv3_lds's image (3x3 convolutions with <= 64 output channels)
        half8* wpk_w = nullptr;       // g_conv3_sww's image (192 -> 64, 3x3: pack_generic_wino)
        float* bias = nullptr;
        int cin_pad = 0, cout_pad = 0;
    };
    std::vector<ConvDev> convs;
    std::vector<float*> prelu;
    std::map<std::tuple<int, int, int>, std::vector<_Float16*>> pool;     // (h, w, channels) -> free zero-bordered arrays
    size_t pool_bytes = 0;
    // g_conv3_sw: the strip segments of an h x w plane, dealt out to `grid` workgroups (sw_segments)
    struct SwPlan { GSwSeg* segs = nullptr; int* seg_begin = nullptr; int grid = 0; };
    std::map<std::vector<int>, SwPlan> sw_plans;                          // (strip columns, h0, w0, h1, w1, ...)
    struct RdbPlan { RdbSeg* segs = nullptr; int* seg_begin = nullptr; int grid = 0; };
    std::map<std::vector<int>, RdbPlan> rdb_plans;                        // (h0, w0, h1, w1, ...)
    std::vector<RdbMatch> rdbs;                                           // find_rdbs() of the loaded graph
    unsigned long long* rdb_dbg = nullptr;                                // UVA_INSTRUMENT + UVA_RDB_STAMPS=1: rdb4_kernel's stamps

    static int pad32(int c) { return (c + 31) / 32 * 32; }

    void release()
    {
        for (auto& c : convs) {
            if (c.wpk) (void)hipFree(c.wpk);
            if (c.wpk_lds) (void)hipFree(c.wpk_lds);
            if (c.wpk_w) (void)hipFree(c.wpk_w);
            if (c.bias) (void)hipFree(c.bias);
        }
        convs.clear();
        for (auto p : prelu)
            if (p) (void)hipFree(p);
        prelu.clear();
        for (auto& kv : pool)
            for (auto p : kv.second) (void)hipFree(p);
        pool.clear();
        pool_bytes = 0;
        for (auto& kv : sw_plans) {
            if (kv.second.segs) (void)hipFree(kv.second.segs);
            if (kv.second.seg_begin) (void)hipFree(kv.second.seg_begin);
        }
        sw_plans.clear();
        for (auto& kv : rdb_plans) {
            if (kv.second.segs) (void)hipFree(kv.second.segs);
            if (kv.second.seg_begin) (void)hipFree(kv.second.seg_begin);
        }
        rdb_plans.clear();
        if (rdb_dbg) (void)hipFree(rdb_dbg);
        rdb_dbg = nullptr;
    }
};

}  // namespace uva
