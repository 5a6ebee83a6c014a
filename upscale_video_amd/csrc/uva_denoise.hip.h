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
// Step 2 is integer arithmetic and bit-exact against the numpy restatement the tests hold; steps 1 and 3 use the CIE formulas in
// fp32 where OpenCV's 8-bit path uses fixed-point tables, so a frame can differ from OpenCV's by an LSB --
// PARITY UNPINNED either way: opencv-python is not installable here.
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>

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

__device__ __forceinline__ float lab_f(float t) { return t > 0.008856f ? cbrtf(t) : 7.787f * t + 16.0f / 116.0f; }

// planes: L [h][w] u8, ab [h][w][2] u8
__global__ void nlm_bgr2lab(const uint8_t* bgr, size_t stride, int h, int w, uint8_t* L, uint8_t* ab)
{
    const int x = blockIdx.x * blockDim.x + threadIdx.x, y = blockIdx.y;
    if (x >= w) return;
    const uint8_t* s = bgr + (size_t)y * stride + (size_t)x * 3;
    const float B = s[0] * (1.0f / 255.0f), G = s[1] * (1.0f / 255.0f), R = s[2] * (1.0f / 255.0f);
    const float X = (0.412453f * R + 0.357580f * G + 0.180423f * B) / 0.950456f;
    const float Y = 0.212671f * R + 0.715160f * G + 0.072169f * B;
    const float Z = (0.019334f * R + 0.119193f * G + 0.950227f * B) / 1.088754f;
    const float fx = lab_f(X), fy = lab_f(Y), fz = lab_f(Z);
    const float Ls = Y > 0.008856f ? 116.0f * fy - 16.0f : 903.3f * Y;
    const float a = 500.0f * (fx - fy), b = 200.0f * (fy - fz);
    const size_t i = (size_t)y * w + x;
    L[i] = (uint8_t)fminf(fmaxf(__builtin_rintf(Ls * 2.55f), 0.f), 255.f);
    ab[2 * i] = (uint8_t)fminf(fmaxf(__builtin_rintf(a + 128.0f), 0.f), 255.f);
    ab[2 * i + 1] = (uint8_t)fminf(fmaxf(__builtin_rintf(b + 128.0f), 0.f), 255.f);
}

__global__ void nlm_lab2bgr(const uint8_t* L, const uint8_t* ab, int h, int w, uint8_t* bgr, size_t stride)
{
    const int x = blockIdx.x * blockDim.x + threadIdx.x, y = blockIdx.y;
    if (x >= w) return;
    const size_t i = (size_t)y * w + x;
    const float Ls = L[i] * (100.0f / 255.0f), a = (float)ab[2 * i] - 128.0f, b = (float)ab[2 * i + 1] - 128.0f;
    float fy, Y;
    if (Ls <= 8.0f) { Y = Ls / 903.3f; fy = 7.787f * Y + 16.0f / 116.0f; }
    else { fy = (Ls + 16.0f) / 116.0f; Y = fy * fy * fy; }
    const float fx = fy + a / 500.0f, fz = fy - b / 200.0f;
    const float ft = 7.787f * 0.008856f + 16.0f / 116.0f;
    const float X = (fx <= ft ? (fx - 16.0f / 116.0f) / 7.787f : fx * fx * fx) * 0.950456f;
    const float Z = (fz <= ft ? (fz - 16.0f / 116.0f) / 7.787f : fz * fz * fz) * 1.088754f;
    const float R = 3.240479f * X - 1.53715f * Y - 0.498535f * Z;
    const float G = -0.969256f * X + 1.875991f * Y + 0.041556f * Z;
    const float B = 0.055648f * X - 0.204043f * Y + 1.057311f * Z;
    uint8_t* d = bgr + (size_t)y * stride + (size_t)x * 3;
    d[0] = (uint8_t)fminf(fmaxf(__builtin_rintf(B * 255.0f), 0.f), 255.f);
    d[1] = (uint8_t)fminf(fmaxf(__builtin_rintf(G * 255.0f), 0.f), 255.f);
    d[2] = (uint8_t)fminf(fmaxf(__builtin_rintf(R * 255.0f), 0.f), 255.f);
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
