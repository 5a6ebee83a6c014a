// uva_sub10.h -- what the host side (uva_api.hip) needs of sub10_kernel (csrc/uva_sub10.hip.h; its own translation unit,
// uva_sub10.hip): constants of the row lists, the argument block and the launcher.
//
// sub10_kernel runs the WHOLE 1x HurrDeblur SubCompact net (models/1x_HurrDeblur_SubCompact_nf24-nc8_244k_net_g.param:3-26:
// conv 3->24, 8 x conv 24->24, conv 24->3, + input; the reference's apply_model, upscale/upscale_processing.py:258-299) in one
// launch, u8 in -> u8 out -- and, since round 6, up to S10_MAXB FRAMES of one geometry in that one launch:
// a frame's strips are dealt out to the 256 workgroups in segments that each pay 20 warm-up rows and the launch pays the
// pipeline's 20 steps of fill and drain once, so k frames in one launch pay both once per k.
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>

#include "uva_devutil.hip.h"

namespace uva {

constexpr int S10_WC = 80;                       // computed columns per strip (five 16-pixel fragments)
constexpr int S10_NL = 10;                       // layers = pipeline stages
constexpr int S10_NW = 12;                       // waves: 8 trunk layers, the first and the last layer on two waves each
constexpr int S10_VALID = S10_WC - 2 * S10_NL;   // columns of the strip the last layer gets right
constexpr int S10_MAX_ROWS = 640;                // row descriptors of a workgroup, copied to LDS (8 B each)
constexpr int S10_DRAIN = 2 * S10_NL;            // steps after the last row went in until it has come out
constexpr int S10_MAXB = 8;                      // frames per launch (uva_net_process_u8_device_batch)
constexpr int S10_YBIAS = 16;                    // a descriptor's row travels as y + S10_YBIAS (rows -10.. are warm-up rows) ...
constexpr int S10_FSHIFT = 16;                   // ... below the frame's index: ((frame << S10_FSHIFT) | (y + S10_YBIAS))
constexpr int S10_MAX_H = (1 << S10_FSHIFT) - 2 * S10_YBIAS;

struct Sub10Args {
    const uint8_t* src[S10_MAXB]; // u8 HWC BGR frames (one plane = a whole frame: apply_model, :263-288), one geometry
    uint8_t* dst[S10_MAXB];
    size_t src_stride;
    size_t dst_stride;
    int h, w;
    const uint4* rows;            // [gridDim.x][max_rows]: x = plane row y (may be outside), y = plane column of computed
                                  // column 0, z = 1: the last layer's row is written out; z >> 8 = the frame; w = rows to the
                                  // nearest row of the segment that is written out
    const int* nrows;             // [gridDim.x]
    int max_rows;
    const half8* wpk[S10_NL];     // pack_sub16 images
    const float* bias[S10_NL];    // [32] each, zero padded
    const float* slope[S10_NL];   // [32] each (none for the last layer)
    unsigned long long* dbg;      // UVA_INSTRUMENT builds: workgroup 0 stamps [step][wave][4] here
};

// One launch of sub10_kernel on `grid` workgroups of 768 threads.  Returns hipSuccess or the failing call's error.
hipError_t launch_sub10_kernel(hipStream_t stream, int grid, const Sub10Args& a);

}  // namespace uva
