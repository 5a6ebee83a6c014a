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

// BinaryOp ADD (ca = cb = 1) and Eltwise SUM with coefficients: whole arrays (0*ca + 0*cb keeps the border zero)
__global__ void g_axpby(const half8* a, float ca, const half8* b, float cb, half8* out, size_t n8)
{
    const size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n8) return;
    const half8 x = a[i], y = b[i];
    half8 r;
#pragma unroll
    for (int e = 0; e < 8; ++e) r[e] = (_Float16)((float)x[e] * ca + (float)y[e] * cb);
    out[i] = r;
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

// ---------------------------------------------------------------------------------------------------
struct GenericDevice {
    struct ConvDev { half8* wpk = nullptr; float* bias = nullptr; int cin_pad = 0, cout_pad = 0; };
    std::vector<ConvDev> convs;
    std::vector<float*> prelu;
    std::map<std::tuple<int, int, int>, std::vector<_Float16*>> pool;     // (h, w, channels) -> free zero-bordered arrays
    size_t pool_bytes = 0;

    static int pad32(int c) { return (c + 31) / 32 * 32; }

    void release()
    {
        for (auto& c : convs) {
            if (c.wpk) (void)hipFree(c.wpk);
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
    }
};

}  // namespace uva
