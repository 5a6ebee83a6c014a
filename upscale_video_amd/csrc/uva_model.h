// uva_model.h -- host-side ncnn .param/.bin loader and MFMA weight packer (no HIP here).
// Replaces ncnn::Net::load_param / load_model as the reference uses them
// (upscale/upscale_processing.py:70-71).
#pragma once
#include <cstdint>
#include <string>
#include <vector>

namespace uva {

struct ConvWeights {
    std::string name;
    int cin = 0, cout = 0;
    int weight_data_size = 0;
    uint32_t tag = 0;             // .bin flag word: 0x01306B47 fp16, 0 fp32
    std::vector<float> w;         // OIHW (ncnn order), fp32 (fp16 expanded exactly)
    std::vector<float> bias;      // [cout]
};

struct Graph {
    bool param_loaded = false, model_loaded = false;
    int scale = 0;                // PixelShuffle factor r (1, 2, 4)
    int nf = 0;                   // trunk width (64 / 24)
    std::vector<ConvWeights> convs;               // head, trunk..., tail
    std::vector<std::string> prelu_names;
    std::vector<std::vector<float>> slopes;       // one per PReLU (convs.size() - 1)
    size_t bin_size = 0, bin_consumed = 0;
};

// Parses the ncnn text graph and accepts only the SRVGGNetCompact pattern.
bool parse_param(const std::string& path, Graph& g, std::string& err);
// Reads the weight stream in graph order; every byte of the file must be consumed.
bool load_bin(const std::string& path, Graph& g, std::string& err);

uint16_t f32_to_f16_bits(float x);   // round-to-nearest-even
float f16_bits_to_f32(uint16_t h);

// MFMA A-operand image of a 3x3 convolution with cin = nf: [KS][MF][64 lanes][8] fp16, where
// lane = (half << 5) | i supplies output channel 32*m + i and K octet ko = 2*ks + half
// (tap = ko / (nf/8), input channels 8*(ko % (nf/8)) .. +7).  Out-of-range -> 0.
void pack_conv3x3(const ConvWeights& c, int nf, std::vector<uint16_t>& out, int* ks_out, int* mf_out);
// The 64-feature trunk kernel uses v_mfma_f32_16x16x32_f16: image [18 k-steps][4 channel blocks][64
// lanes][8] fp16, k-step = 2*tap + ch; lane = (octet << 4) | i supplies output channel 16*mb + i and
// input channels 32*ch + 8*octet .. +7 of that tap.
void pack_trunk64(const ConvWeights& c, std::vector<uint16_t>& out);
// trunkw_kernel (csrc/uva_wino.hip.h): the same 64 -> 64 convolution as 1-D Winograd F(2,3) along x.  Image
// [j 0..3][dy 0..2][ch 0..1][4 channel blocks][64 lanes][8] fp16 of the TRANSFORMED taps of row dy
// (g0, g1, g2 = that row's three taps, taken at full precision; one rounding to fp16 at the end):
//   U0 = g0   U1 = (g0 + g1 + g2) / 2   U2 = (g0 - g1 + g2) / 2   U3 = g2
// lanes as in pack_trunk64.  The CPU checker (conv2d_wino_f23, test side) spells the same expressions.
// in_sign[64] / out_sign[64] (+1 / -1, may be null): the layer computed on negated input channels / computing negated
// output channels -- how trunkw_kernel's packed-fp16 PReLU max(x, slope * x) serves channels whose slope exceeds 1
// (max(-x, -slope * x) = -PReLU(x), every step exact; see pack_sub16).
void pack_trunk64_wino(const ConvWeights& c, std::vector<uint16_t>& out, const float* in_sign = nullptr, const float* out_sign = nullptr);
// tail_kernel<64, R> (v_mfma_f32_16x16x32_f16, cin = 64, cout = 3*R*R padded to a multiple of 16):
// image [18 k-steps][MB = ceil(cout/16) blocks][64 lanes][8] fp16, lanes and k-steps as pack_trunk64.
void pack_tail64(const ConvWeights& c, std::vector<uint16_t>& out);
// Head (cin = 3): K = [tap][4] (3 channels + zero), octet o = 2*ks + half holds taps 2o, 2o+1;
// image [3][MF][64][8].
void pack_head(const ConvWeights& c, std::vector<uint16_t>& out, int* mf_out);

// sub10_kernel (the whole 24-feature 1x net in one launch, v_mfma_f32_16x16x32_f16): image [k-step][m-block][64 lanes][8],
// lane = (o << 4) | i supplies output row i of block mb and the K octet of slot (ks, o):
//   cin = 24 (trunk, tail): octet id = 3*tap + channel octet, 27 octets in 7 k-steps as SUB16_OCTET[ks][o] (27 = none);
//   cin = 3  (head):        slot 4ks + o holds taps 2(4ks+o) and 2(4ks+o)+1 as [B, G, R, 0] each, 5 octets -> 2 k-steps;
//   cout = 24: block 1 holds channels 16..23 in rows 4q, 4q+1 (channel 16 + 2q + j), rows 4q+2, 4q+3 are zero.
// SUB16_OCTET pairs, in the lanes o = 0,1 and o = 2,3 of a k-step, octets whose (column + channel octet) parity is the
// same: the kernel's 16-byte LDS reads of 48-byte pixels are then conflict-free (csrc/uva_kernels.hip.h sub10_body).
// in_sign[cin] / out_sign[cout] (+-1, may be null) flip the sign of input / output channels: the kernel computes PReLU
// as max(x, slope*x), which is PReLU only for slope <= 1 -- a channel with a larger slope is computed negated
// (max(-x, -slope*x) = -PReLU(x), every step exact) and the next layer's weights take the sign back.
constexpr unsigned char SUB16_OCTET[7][4] = {{0, 2, 4, 6},     {8, 9, 11, 13},   {15, 17, 18, 20}, {22, 24, 26, 27},
                                             {1, 3, 5, 7},     {10, 12, 14, 16}, {19, 21, 23, 25}};
void pack_sub16(const ConvWeights& c, std::vector<uint16_t>& out, int* ks_out, int* mb_out, const float* in_sign = nullptr,
                const float* out_sign = nullptr);

}  // namespace uva
