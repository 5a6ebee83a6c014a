// uva_sub5.h -- what the host side (uva_api.hip) needs of sub5_kernel (csrc/uva_sub5.hip.h; its own translation unit,
// uva_sub5.hip): constants of the row lists, the argument block and the launcher.  No device code here.
//
// sub5_kernel<PART> runs the 1x HurrDeblur SubCompact net (models/1x_HurrDeblur_SubCompact_nf24-nc8_244k_net_g.param:3-26:
// conv 3->24, 8 x conv 24->24, conv 24->3, + input) as TWO launches of FIVE layers each, two five-layer pipelines (two
// neighbouring strips) per workgroup -- the variant of sub10_kernel that removes WORK: 54 of 64 computed columns and
// n of n + 10 rows are kept per launch where sub10_kernel keeps 60 of 80 and n of n + 20.
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>

namespace uva {

constexpr int S5_WC = 64;                        // computed columns per strip: four 16-pixel MFMA fragments
constexpr int S5_NL = 5;                         // layers per launch = stages of a pipeline
constexpr int S5_VALID = S5_WC - 2 * S5_NL;      // 54: the columns of a strip the fifth layer gets right
constexpr int S5_PAIRW = 2 * S5_VALID;           // 108: a workgroup's two pipelines cover neighbouring strips
constexpr int S5_MAX_ROWS = 1024;                // row descriptors of a workgroup, copied to LDS (8 B each)
constexpr int S5_MIDB = 48;                      // bytes per pixel of the 24-channel image between the two launches (fp16, ring order)

struct Sub5Args {
    const uint8_t* src;           // u8 HWC BGR frame (part 0: the input rows; part 1: the residual)
    size_t src_stride;
    uint8_t* dst;                 // u8 HWC BGR result (part 1)
    size_t dst_stride;
    char* mid;                    // [h][w] pixels of S5_MIDB bytes: layer 4's output, written by part 0 and read by part 1
    int h, w;
    const uint4* rows;            // [grid][max_rows]: x = plane row y (may be outside), y = plane column of computed column 0 of the
                                  // FIRST pipeline (the second one's is S5_VALID further right), z = 1: the row is written out
    const int* nrows;             // [grid]
    int max_rows;
    const void* wpk[S5_NL];       // pack_sub16 images of this launch's five layers (part 0: layers 0..4, part 1: layers 5..9)
    const float* bias[S5_NL];     // [32] each, zero padded
    const float* slope[S5_NL];    // [32] each (unused for the net's last layer)
    unsigned long long* dbg;      // UVA_INSTRUMENT builds: workgroup 0 stamps [step][wave][4]
};

// One launch of sub5_kernel<part> on `grid` workgroups of 768 threads.  Returns hipSuccess or the failing call's error.
hipError_t launch_sub5_kernel(hipStream_t stream, int grid, const Sub5Args& a, int part);

}  // namespace uva
