// uva_denoise.hip.h -- `-m n=K` on the MI355X: the reference's
//     cv2.fastNlMeansDenoisingColored(img, None, K, K, 5, 9)          upscale/upscale_processing.py:350-361
// restated from OpenCV's published algorithm (modules/photo/src/denoising.cpp fastNlMeansDenoisingColored,
// fast_nlmeans_denoising_invoker.hpp / _commons.hpp):
//   1. BGR -> 8-bit CIE Lab, the frame taken as LINEAR light (COLOR_LBGR2Lab: no sRGB curve);
//   2. non-local means on the L plane with h, and on the 2-channel (a, b) image with hColor: template 5x5,
//      search 9x9, reflect-101 border of 6 pixels, squared-difference patch distance summed over the template
//      and the channels, `>> 5` (25 template pixels rounded up to 32), an integer weight table
//      round(F * exp(-(d * 32/25) / (h*h*channels))) with F = INT_MAX / (81 * 255) and weights below F/1000
//      dropped, integer accumulation, rounded division by the weight sum;
//   3. Lab -> BGR (COLOR_Lab2LBGR).
// All three steps are integer arithmetic (steps 1 and 3: OpenCV's fixed-point table code for 8-bit Lab, not the CIE
// formulas) and bit-exact, end to end, against the numpy restatement the tests hold.  PARITY UNPINNED all the same:
// opencv-python is not installable here, the table code is restated from memory, and the reference passes a cv2.UMat,
// i.e. takes OpenCV's OpenCL branch where there is one.
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>

#include <cmath>

namespace uva {

constexpr int NLM_T = 2;                 // template half size (5x5)
constexpr int NLM_S = 4;                 // search half size (9x9)
constexpr int NLM_B = NLM_T + NLM_S;     // border (6)
constexpr int NLM_SHIFT = 5;             // 25 template pixels -> next power of two
constexpr int NLM_BLK = 16;              // output pixels per workgroup side
constexpr int NLM_TILE = NLM_BLK + 2 * NLM_B;

__device__ __forceinline__ int reflect101(int i, int n)
{
    if (n == 1) return 0;
    while (i < 0 || i >= n) i = i < 0 ? -i : 2 * n - 2 - i;      // cv::BORDER_DEFAULT
    return i;
}

// ---- OpenCV's 8-bit Lab conversions: fixed-point table code, not the CIE formulas (modules/imgproc/src/color_lab.cpp:
// RGB2Lab_b, Lab2RGBinteger, initLabTabs), restated from memory of the 4.x source -- the tests hold the same restatement in numpy
// and DESIGN.md section 5.6 says what that is worth (PARITY UNPINNED).  Constants: lab_shift 12, gamma_shift 3,
// lab_shift2 15, base_shift 14, inv_gamma_shift 12.
constexpr int LAB_CBRT_TAB_SIZE_B = 256 * 3 / 2 * 8;
struct LabTables {
    int cbrt_tab[LAB_CBRT_TAB_SIZE_B];   // LabCbrtTab_b: round(32768 * f(i / (255 * 8)))
    int fwd[9];                          // (X/Xn, Y, Z/Zn) from (R, G, B), << 12
    int inv[9];                          // (R, G, B) from (x, y, z) with the white point folded in, << 12
    int y_of_l[256], fy_of_l[256];       // LabToYF_b
};
inline void lab_tables_host(LabTables& t)
{
    // OpenCV builds these in softfloat / softdouble (IEEE, round to nearest even): plain float / double arithmetic, one
    // operation per statement; mulAdd is a fused multiply-add (one rounding: through double); cv::cbrt -> libm's
    const float scale = 1.0f / (255.0f * 8.0f), lthresh = 216.0f / 24389.0f, lscale = 841.0f / 108.0f, lbias = 16.0f / 116.0f;
    for (int i = 0; i < LAB_CBRT_TAB_SIZE_B; ++i) {
        const float x = scale * (float)i;
        const float v = x < lthresh ? (float)((double)x * (double)lscale + (double)lbias) : (float)std::cbrt((double)x);
        t.cbrt_tab[i] = (int)std::lrintf(32768.0f * v);
    }
    static const double rgb2xyz[9] = {0.412453, 0.357580, 0.180423, 0.212671, 0.715160, 0.072169, 0.019334, 0.119193, 0.950227};
    static const double xyz2rgb[9] = {3.240479, -1.53715, -0.498535, -0.969256, 1.875991, 0.041556, 0.055648, -0.204043, 1.057311};
    static const double d65[3] = {0.950456, 1.0, 1.088754};
    for (int i = 0; i < 3; ++i)
        for (int j = 0; j < 3; ++j) {
            t.fwd[i * 3 + j] = (int)std::lrint(4096.0 * rgb2xyz[i * 3 + j] / d65[i]);
            t.inv[i * 3 + j] = (int)std::lrint(4096.0 * xyz2rgb[i * 3 + j] * d65[j]);
        }
    const int base = 1 << 14;
    for (int i = 0; i < 256; ++i) {
        if (i <= 20) {
            const float a = (float)(i * base * 20 * 9), b = (float)(17 * 29 * 29 * 29);
            t.y_of_l[i] = (int)std::lrintf(a / b);
            const float c = 16.0f / 116.0f, d = (float)(i * 5) / (float)(3 * 17 * 29), e = c + d;
            t.fy_of_l[i] = (int)std::lrintf((float)base * e);
        } else {
            const float a = (float)(i * 100 * base) / (float)(255 * 116), b = (float)(16 * base) / 116.0f, fy = a + b;
            t.fy_of_l[i] = (int)std::lrintf(fy);
            const float f2 = fy * fy, f3 = f2 * fy;
            t.y_of_l[i] = (int)std::lrintf(f3 / (float)(base * base));
        }
    }
}
__device__ __forceinline__ int lab_descale(int x, int n) { return (x + (1 << (n - 1))) >> n; }
__device__ __forceinline__ int lab_sat8(int v) { return v < 0 ? 0 : v > 255 ? 255 : v; }
// abToXZ_b[v - minABvalue]: C integer arithmetic (division towards zero), computed instead of looked up
__device__ __forceinline__ int lab_ab_to_xz(int v)
{
    if (v <= 3390) return v * 108 / 841 - (16384 * 16 / 116 * 108 / 841);
    return (int)(((long long)v * v / 16384) * v / 16384);
}

// planes: L [h][w] u8, ab [h][w][2] u8.  COLOR_LBGR2Lab: the frame is taken as LINEAR light (gamma table i << 3)
__global__ void nlm_bgr2lab(const uint8_t* bgr, size_t stride, int h, int w, uint8_t* L, uint8_t* ab, const LabTables* t)
{
    const int x = blockIdx.x * blockDim.x + threadIdx.x, y = blockIdx.y;
    if (x >= w) return;
    const uint8_t* s = bgr + (size_t)y * stride + (size_t)x * 3;
    const int B = s[0] << 3, G = s[1] << 3, R = s[2] << 3;
    const int fX = t->cbrt_tab[lab_descale(R * t->fwd[0] + G * t->fwd[1] + B * t->fwd[2], 12)];
    const int fY = t->cbrt_tab[lab_descale(R * t->fwd[3] + G * t->fwd[4] + B * t->fwd[5], 12)];
    const int fZ = t->cbrt_tab[lab_descale(R * t->fwd[6] + G * t->fwd[7] + B * t->fwd[8], 12)];
    constexpr int Lscale = (116 * 255 + 50) / 100, Lshift = -((16 * 255 * (1 << 15) + 50) / 100);
    const size_t i = (size_t)y * w + x;
    L[i] = (uint8_t)lab_sat8(lab_descale(Lscale * fY + Lshift, 15));
    ab[2 * i] = (uint8_t)lab_sat8(lab_descale(500 * (fX - fY) + 128 * (1 << 15), 15));
    ab[2 * i + 1] = (uint8_t)lab_sat8(lab_descale(200 * (fY - fZ) + 128 * (1 << 15), 15));
}

// COLOR_Lab2LBGR (Lab2RGBinteger; linear inverse gamma table (v * 255) >> 12)
__global__ void nlm_lab2bgr(const uint8_t* L, const uint8_t* ab, int h, int w, uint8_t* bgr, size_t stride, const LabTables* t)
{
    const int x = blockIdx.x * blockDim.x + threadIdx.x, y = blockIdx.y;
    if (x >= w) return;
    const size_t i = (size_t)y * w + x;
    const int LL = L[i], aa = ab[2 * i], bb = ab[2 * i + 1];
    const int yy = t->y_of_l[LL], ify = t->fy_of_l[LL];
    const int adiv = ((5 * aa * 53687 + (1 << 7)) >> 13) - 128 * 16384 / 500;
    const int bdiv = ((bb * 41943 + (1 << 4)) >> 9) - 128 * 16384 / 200 + 1;
    const int xx = lab_ab_to_xz(ify + adiv), zz = lab_ab_to_xz(ify - bdiv);
    uint8_t* d = bgr + (size_t)y * stride + (size_t)x * 3;
#pragma unroll
    for (int k = 0; k < 3; ++k) {        // R, G, B rows of the matrix -> bytes 2, 1, 0
        int v = lab_descale(t->inv[3 * k] * xx + t->inv[3 * k + 1] * yy + t->inv[3 * k + 2] * zz, 14);
        v = v < 0 ? 0 : v > 4095 ? 4095 : v;
        d[2 - k] = (uint8_t)((v * 255) >> 12);
    }
}

// One workgroup = 16 x 16 output pixels; their 28 x 28 neighbourhood (reflect-101 at the image edge) is staged in
// LDS once.  The patch distance of a search offset is a 5 x 5 box sum of per-pixel squared differences, and the boxes of
// neighbouring pixels overlap: per offset every thread adds up ONE column of five differences (320 column sums for the
// 16 x 20 positions the block needs, through a double-buffered LDS array, one barrier per offset) and then five of its
// neighbours' column sums -- 6.25 squared differences per pixel and offset instead of 25, the same integers.  HBM
// traffic is one read and one write of the plane.
template <int CN>
__global__ __launch_bounds__(NLM_BLK * NLM_BLK) void nlm_plane(const uint8_t* src, int h, int w, const int* weight_table,
                                                                int table_size, uint8_t* dst)
{
    constexpr int VW = NLM_BLK + 2 * NLM_T;                       // 20 column sums per output row
    __shared__ uint8_t tile[NLM_TILE * NLM_TILE * CN];
    __shared__ int vsum[2][NLM_BLK * VW];
    const int bx = blockIdx.x * NLM_BLK, by = blockIdx.y * NLM_BLK;
    for (int i = threadIdx.x; i < NLM_TILE * NLM_TILE; i += NLM_BLK * NLM_BLK) {
        const int ty = i / NLM_TILE, tx = i - ty * NLM_TILE;
        const int sy = reflect101(by + ty - NLM_B, h), sx = reflect101(bx + tx - NLM_B, w);
#pragma unroll
        for (int c = 0; c < CN; ++c) tile[i * CN + c] = src[((size_t)sy * w + sx) * CN + c];
    }
    __syncthreads();
    const int lx = threadIdx.x % NLM_BLK, ly = threadIdx.x / NLM_BLK;
    const int x = bx + lx, y = by + ly;
    const int cx = lx + NLM_B, cy = ly + NLM_B;
    // column sum (r, c): rows cy(r) - 2 .. + 2 of tile column NLM_B - 2 + c
    auto colsum = [&](int r, int c, int sy, int sx) {
        const int tx = NLM_B - NLM_T + c;
        int v = 0;
#pragma unroll
        for (int ty = -NLM_T; ty <= NLM_T; ++ty) {
            const uint8_t* a = tile + ((r + NLM_B + ty) * NLM_TILE + tx) * CN;
            const uint8_t* b = tile + ((r + NLM_B + ty + sy) * NLM_TILE + tx + sx) * CN;
#pragma unroll
            for (int c2 = 0; c2 < CN; ++c2) {
                const int d = (int)a[c2] - (int)b[c2];
                v += d * d;
            }
        }
        return v;
    };
    int est[CN];
#pragma unroll
    for (int c = 0; c < CN; ++c) est[c] = 0;
    int wsum = 0, buf = 0;
    for (int sy = -NLM_S; sy <= NLM_S; ++sy)
        for (int sx = -NLM_S; sx <= NLM_S; ++sx) {
            int* const vb = vsum[buf];
            vb[ly * VW + lx] = colsum(ly, lx, sy, sx);
            if (threadIdx.x < NLM_BLK * 2 * NLM_T) {              // the four extra columns of the 16 rows
                const int r = threadIdx.x / (2 * NLM_T), c = NLM_BLK + threadIdx.x % (2 * NLM_T);
                vb[r * VW + c] = colsum(r, c, sy, sx);
            }
            __syncthreads();                                      // (the other buffer is free again after the NEXT barrier)
            int dist = 0;
#pragma unroll
            for (int t = 0; t <= 2 * NLM_T; ++t) dist += vb[ly * VW + lx + t];
            const int idx = min(dist >> NLM_SHIFT, table_size - 1);
            const int wgt = weight_table[idx];
            const uint8_t* p = tile + ((cy + sy) * NLM_TILE + cx + sx) * CN;
#pragma unroll
            for (int c = 0; c < CN; ++c) est[c] += wgt * (int)p[c];
            wsum += wgt;
            buf ^= 1;
        }
    if (x >= w || y >= h) return;
#pragma unroll
    for (int c = 0; c < CN; ++c) {
        const unsigned q = ((unsigned)est[c] + (unsigned)wsum / 2) / (unsigned)wsum;     // divByWeightsSum
        dst[((size_t)y * w + x) * CN + c] = (uint8_t)min(q, 255u);
    }
}

}  // namespace uva
