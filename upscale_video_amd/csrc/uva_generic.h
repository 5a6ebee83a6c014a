// uva_generic.h -- generic ncnn .param graphs beyond the SRVGGNetCompact pattern (SURVEY.md section 8f
// rank 3): what `-m r` selects in the reference, models/4x_Valar_v1.param:3-1208 (an ESRGAN+ RRDB net,
// upscale/upscale_processing.py:913-916).  Layer types understood, each with ncnn's published semantics:
//   Input; Split (aliases); Convolution 3x3 pad 1 / 1x1 pad 0, stride 1, optional bias, optional fused
//   LeakyReLU (activation_type 9=2, -23310=1,slope); Concat along channels; BinaryOp ADD; Eltwise SUM with
//   coefficients (0=1 -23301=n,c0,c1..); Interp nearest with integer scale; PReLU; PixelShuffle.
// Host side here (parser, shape inference, .bin reader); the device executor is uva_generic.hip.
// The Valar weights are a missing blob upstream (.MISSING_LARGE_BLOBS): the graph runs with whatever
// .bin matches it, the tests use synthetic weights.
#pragma once
#include <string>
#include <vector>

#include "uva_model.h"

namespace uva {

struct GLayer {
    enum Kind { INPUT, SPLIT, CONV, CONCAT, ADD, ELTWISE_SUM, INTERP_NEAREST, PRELU, PIXELSHUFFLE };
    Kind kind = INPUT;
    std::string name;
    std::vector<int> in, out;     // blob ids
    // CONV
    int conv = -1;                // index into GenericGraph::convs
    int ksize = 3;                // 1 or 3
    bool has_bias = false;
    bool has_act = false;         // fused LeakyReLU
    float act_slope = 0.f;
    // ELTWISE_SUM
    std::vector<float> coeffs;
    // INTERP_NEAREST / PIXELSHUFFLE
    int factor = 1;
    // PRELU
    int slopes = -1;              // index into GenericGraph::prelu
    // CONCAT: 0 copy every input; 1 first Concat of a dense chain: only input 0 is copied (into the chain's shared
    // array), 2 later Concat of the chain: nothing to do (plan_concat_groups)
    int concat_mode = 0;
};

struct GBlob {
    std::string name;
    int channels = 0;
    int scale = 1;                // spatial size relative to the input
    int alias_of = -1;            // Split outputs alias their input
    int consumers = 0;
    // dense-chain plan (plan_concat_groups): the blob lives in channels [group_off, group_off + channels) of the shared
    // array of group `group`
    int group = -1, group_off = 0;
};

struct GenericGraph {
    bool param_loaded = false, model_loaded = false;
    std::vector<GLayer> layers;
    std::vector<GBlob> blobs;
    std::vector<ConvWeights> convs;
    std::vector<std::vector<float>> prelu;
    std::vector<int> prelu_sizes;
    int in_blob = -1, out_blob = -1;
    int scale = 1;                // of the output blob
    int max_channels = 0;
    double flops_per_input_px = 0;
    std::vector<int> group_channels;      // per dense chain: channels of its shared array
    std::vector<int> group_blobs;         // ... and how many blobs live in it
};

// Dense blocks (RRDB: x1 = conv(x), x2 = conv(cat(x, x1)), x3 = conv(cat(x, x1, x2)), ...) concatenate a growing
// prefix over and over.  Where every Concat of such a chain extends the previous one by one convolution output, the
// outputs are only read by convolutions that can read a channel prefix of a wider array (g_conv3_lds) and the new
// members are only read by the chain's Concats, the whole chain shares ONE array: convolutions write their output
// into their channel range, the first Concat copies x, the others cost nothing.  Fills GLayer::concat_mode,
// GBlob::group / group_off, GenericGraph::group_*.  Pure analysis: the executor may ignore it.
void plan_concat_groups(GenericGraph& g);

// A residual dense block whose first four convolutions rdb4_kernel (csrc/uva_rdb.hip.h) runs in one launch:
//   x1 = lrelu(conv3(x)), x2 = lrelu(conv3(x,x1)) + conv1(x), x3 = lrelu(conv3(x,x1,x2)), x4 = lrelu(conv3(x,x1,x2,x3)) + x2
// (models/4x_Valar_v1.param:6-19), all five blobs members of one dense chain's 192-channel array (plan_concat_groups)
// with x already in it.  Layer indices; `skip` = the layers the launch at c1 replaces.
struct RdbMatch {
    int group = -1;
    int c1 = -1, c2 = -1, c2s = -1, add2 = -1, c3 = -1, c4 = -1, add4 = -1;
    float slope = 0.f;
};
std::vector<RdbMatch> find_rdbs(const GenericGraph& g);

bool parse_param_generic(const std::string& path, GenericGraph& g, std::string& err);
bool load_bin_generic(const std::string& path, GenericGraph& g, std::string& err);

// MFMA A-operand image of a generic convolution for v_mfma_f32_16x16x32_f16:
// [tap][cin_pad/32][cout_pad/16][64 lanes][8] fp16; lane = (octet << 4) | i supplies output channel
// 16*mb + i and input channels 32*c32 + 8*octet .. +7 of that tap; out-of-range -> 0.
// lds_order: the image g_conv3_lds reads -- the lane groups (lane >> 4) = 0..3 hold the octets {0, 2, 1, 3} of the 32.
void pack_generic(const ConvWeights& c, int ksize, int cin_pad, int cout_pad, std::vector<uint16_t>& out, bool lds_order = false);
void pack_generic_wino(const ConvWeights& c, int cin_pad, int cout_pad, std::vector<uint16_t>& out);     // 3x3 only: g_conv3_sww's image

}  // namespace uva
