// uva_wino.h -- what the host side (uva_api.hip) needs of trunkw_kernel (csrc/uva_wino.hip.h; its own translation
// unit, uva_wino.hip): the step-list constants, the argument block and the launcher.  No device code here.
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>

namespace uva {

struct Trunk2Step;                                // uva_devutil.hip.h (32 bytes; trunkw's meaning of the fields: uva_wino.hip.h)

constexpr int TW_SW = 30;                         // output columns per strip
constexpr int TW_PAD_STEPS = 2;                   // dummy entries behind a workgroup's last step (DMA look-ahead)
constexpr int TW_FOLD_MAXW = 12;                  // widest last strip two planes can share (folded steps: pairs 0..6 per plane)

struct TrunkwArgs {
    const char* in_act;           // activation buffer INCLUDING its leading guard
    char* out_act;
    const void* wpk[2];           // pack_trunk64_wino images of layer i and i+1
    const float* bias[2];
    const float* slope[2];
    const Trunk2Step* steps;      // [grid][max_steps + TW_PAD_STEPS]
    const int* nsteps;            // [grid]
    int max_steps;
    _Float16* sink;
    unsigned long long* dbg;      // instrumented builds (-DUVA_INSTRUMENT): s_memtime stamps of workgroup 0, wave 0 of each group:
                                  // [16 * it + 8 * group + {0: iteration start, 1: phase X work done, 2: barrier 1 passed,
                                  // 3: phase Y work done, 4: (A) the epilogue's ring writes done}], entry time at [16 * niter]
};

// How a launch applies PReLU (bias and slopes of both layers as float arrays either way):
//   TW_ACT_F32     on the fp32 sums, v_med3_f32(x, slope * x, +-inf), then one rounding to fp16 (trunk2_kernel's way)
//   TW_ACT_F16     sums rounded to fp16 first, then max(x, slope16 * x) on packed halves (4 instructions fewer per block row
//                  and group).  That IS PReLU for slope <= 1; a channel with a larger slope must arrive NEGATED (weights and
//                  bias of its layer packed with out_sign -1, the next layer's with in_sign -1: uva_model.h) ...
//   TW_ACT_F16_FLIP ... which for the FIRST layer of a launch stays inside the launch; the second layer's negated channels
//                  (those whose slope[1] exceeds 1) get their sign back in front of the stores to HBM.
enum { TW_ACT_F32 = 0, TW_ACT_F16 = 1, TW_ACT_F16_FLIP = 2 };

// One launch of trunkw_kernel<64, act> on `grid` workgroups.  Returns hipSuccess or the failing call's error.
hipError_t launch_trunkw_kernel(hipStream_t stream, int grid, const TrunkwArgs& a, int act);

}  // namespace uva
