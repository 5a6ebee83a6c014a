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
// tail_kernel<64, R> (v_mfma_f32_16x16x32_f16, cin = 64, cout = 3*R*R padded to a multiple of 16):
// image [18 k-steps][MB = ceil(cout/16) blocks][64 lanes][8] fp16, lanes and k-steps as pack_trunk64.
void pack_tail64(const ConvWeights& c, std::vector<uint16_t>& out);
// Head (cin = 3): K = [tap][4] (3 channels + zero), octet o = 2*ks + half holds taps 2o, 2o+1;
// image [3][MF][64][8].
void pack_head(const ConvWeights& c, std::vector<uint16_t>& out, int* mf_out);

// sub10_kernel (the whole 24-feature 1x net in one launch, v_mfma_f32_16x16x32_f16): image [k-step][m-block][64 lanes][8],
// lane = (o << 4) | i supplies output channel 16*mb + i and K octet ko = 4*ks + o.
//   cin = 24 (trunk, tail): ko = 3*tap + channel octet, 27 octets -> 7 k-steps; cout padded to 16*mbn;
//   cin = 3  (head):        octet ko holds taps 2ko and 2ko+1 as [B, G, R, 0] each, 5 octets -> 2 k-steps.
void pack_sub16(const ConvWeights& c, std::vector<uint16_t>& out, int* ks_out, int* mb_out);

}  // namespace uva
