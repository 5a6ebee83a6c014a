// uva_sw.h -- what g_conv3_sw (csrc/uva_rdb.hip.h) and g_conv3_sww (csrc/uva_sww.hip.h, a translation unit of its own) share:
// the segment and argument structs of the strip-walking convolutions of 4x_Valar_v1 (`-m r`, models/4x_Valar_v1.param) and the
// one spelling of an element-wise sum.
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>

#include "uva_devutil.hip.h"

namespace uva {

struct GSwSeg { int c0, y0, y1, plane; };     // output columns [c0, c0 + SW_C) x rows [y0, y1) of plane `plane`, y1 - y0 a multiple
                                              // of 4 except at the plane's bottom
constexpr int GEN_MAX_PLANES = 16;            // planes (reference tiles) of a frame that one launch of the kernels below takes

struct GSwArgs {
    // per plane (all planes of a launch: same layer, same strides): zero-bordered arrays [(h+3)][(w+2)][stride], pixel (y, x)
    // at row y+1, column x+1
    const _Float16* in[GEN_MAX_PLANES];     // channels read: [0, 32*KC)
    _Float16* out[GEN_MAX_PLANES];
    const _Float16* res[GEN_MAX_PLANES];    // the element-wise sum behind the convolution (GConvArgs::res: same expression, same rounding)
    const _Float16* res2[GEN_MAX_PLANES];   // ... and a second sum behind the first: out = x2*ca2 + y2*cb2, one of them the first sum's result
    int ph[GEN_MAX_PLANES], pw[GEN_MAX_PLANES];
    int in_stride;                // elements per pixel
    const half8* wpk;             // pack_generic image (natural octet order): [tap][KC][4][64 lanes][8]
    const float* bias;            // [64]
    int out_stride, out_coff;
    float slope;                  // LeakyReLU (template ACT)
    int res_stride;
    float ca, cb;
    int res2_stride;
    float ca2, cb2;
    const GSwSeg* segs;           // this launch's segments; workgroup g owns segs[seg_begin[g] .. seg_begin[g+1])
    const int* seg_begin;
    _Float16* sink;               // >= 64 * 8 bytes: where lanes outside the plane store to
};

constexpr int SW_R = 4;                       // output rows per block

// out = x*ca + y*cb of an element-wise sum (BinaryOp ADD, Eltwise SUM with coefficients), rounded to fp16: ONE spelling
// for every kernel that computes it (g_axpby, g_axpby_strided and the convolution epilogues that absorb a sum), so that a
// sum gives the same bytes whichever kernel does it -- left to the compiler, `x*ca + y*cb` contracts into an fma around
// either product
__device__ __forceinline__ _Float16 g_axpby1(float x, float ca, float y, float cb)
{
    // The fp32 result is made opaque before it is rounded to fp16: where the compiler sees both steps it may pick
    // v_fma_mixlo_f16, which rounds the exact fma ONCE -- other call sites round twice (fp32, then fp16), and the two
    // differ in rare ties.
    float t = __builtin_fmaf(x, ca, y * cb);
    asm volatile("" : "+v"(t));
    return (_Float16)t;
}

// g_conv3_sww<RES, RES2> (192 -> 64 as 1-D Winograd F(2,3), csrc/uva_sww.hip.h): one launch on `grid` workgroups of 256 threads;
// a.wpk = pack_generic_wino's image.  res / res2 as g_conv3_sw's RES / RES2 (0 none, 1 the sum's x is the other operand, 2 the
// convolution's side is); the instantiations that exist are (0,0), (2,0), (2,2) -- what 4x_Valar_v1 needs.
hipError_t launch_conv3_sww(hipStream_t stream, int grid, const GSwArgs& a, int res, int res2);

}  // namespace uva
