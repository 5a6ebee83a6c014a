// uva_model.cpp -- ncnn .param/.bin loader restricted to the SRVGGNetCompact pattern, and the
// repacking of OIHW weights into MFMA A-operand order.  Host only.
//
// File formats follow ncnn's published readers (src/net.cpp load_param text format, magic
// 7767517; src/modelbin.cpp: Convolution weights carry a u32 flag -- 0x01306B47 = fp16 payload
// padded to 4 bytes, 0 = raw fp32 -- while bias and PReLU slopes are raw fp32).  The reference
// loads them at upscale/upscale_processing.py:70-71.
#include "uva_model.h"

#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <fstream>
#include <map>
#include <sstream>

namespace uva {

float f16_bits_to_f32(uint16_t h)
{
    const uint32_t sign = (uint32_t)(h & 0x8000u) << 16;
    uint32_t exp = (h >> 10) & 0x1f, man = h & 0x3ffu, f;
    if (exp == 0) {
        if (man == 0) f = sign;
        else {
            int e = -1;
            do { man <<= 1; ++e; } while (!(man & 0x400u));
            f = sign | ((uint32_t)(127 - 15 - e) << 23) | ((man & 0x3ffu) << 13);
        }
    } else if (exp == 31) f = sign | 0x7f800000u | (man << 13);
    else f = sign | ((exp + 112) << 23) | (man << 13);
    float out;
    std::memcpy(&out, &f, 4);
    return out;
}

uint16_t f32_to_f16_bits(float x)
{
    uint32_t f;
    std::memcpy(&f, &x, 4);
    const uint32_t sign = (f >> 16) & 0x8000u, a = f & 0x7fffffffu;
    if (a >= 0x7f800000u) return (uint16_t)(sign | 0x7c00u | (a > 0x7f800000u ? 0x200u : 0));
    if (a >= 0x477ff000u) return (uint16_t)(sign | 0x7c00u);
    if (a < 0x33000001u) return (uint16_t)sign;
    const int e = (int)(a >> 23) - 127;
    const uint32_t m = (a & 0x7fffffu) | 0x800000u;
    const int shift = e < -14 ? 13 + (-14 - e) : 13;
    uint32_t q = m >> shift;
    const uint32_t rem = m & ((1u << shift) - 1), halfway = 1u << (shift - 1);
    if (rem > halfway || (rem == halfway && (q & 1))) ++q;
    const uint32_t h = e < -14 ? q : ((uint32_t)(e + 14) << 10) + q;
    return (uint16_t)(sign | h);
}

namespace {

struct RawLayer {
    std::string type, name;
    std::vector<std::string> in, out;
    std::map<int, std::string> kv;
    int geti(int id, int def) const { auto it = kv.find(id); return it == kv.end() ? def : std::atoi(it->second.c_str()); }
    double getf(int id, double def) const { auto it = kv.find(id); return it == kv.end() ? def : std::atof(it->second.c_str()); }
};

bool fail(std::string& err, const std::string& msg) { err = msg; return false; }

}  // namespace

bool parse_param(const std::string& path, Graph& g, std::string& err)
{
    g = Graph();
    std::ifstream f(path);
    if (!f) return fail(err, "load_param: cannot open " + path);
    long magic = 0;
    int nl = 0, nb = 0;
    f >> magic >> nl >> nb;
    if (!f || magic != 7767517) return fail(err, "load_param: bad magic in " + path);
    if (nl < 8 || nl > 4096) return fail(err, "load_param: bad layer count in " + path);
    std::string line;
    std::getline(f, line);
    std::vector<RawLayer> L;
    while ((int)L.size() < nl && std::getline(f, line)) {
        std::istringstream ss(line);
        RawLayer r;
        int nin = 0, nout = 0;
        if (!(ss >> r.type >> r.name >> nin >> nout)) continue;
        std::string t;
        for (int i = 0; i < nin; ++i) { ss >> t; r.in.push_back(t); }
        for (int i = 0; i < nout; ++i) { ss >> t; r.out.push_back(t); }
        while (ss >> t) {
            const size_t eq = t.find('=');
            if (eq == std::string::npos) continue;
            r.kv[std::atoi(t.substr(0, eq).c_str())] = t.substr(eq + 1);
        }
        L.push_back(r);
    }
    if ((int)L.size() != nl) return fail(err, "load_param: truncated " + path);

    // SRVGGNetCompact: Input, Split, Conv(3->nf), [PReLU, Conv]*, PixelShuffle r,
    // Interp nearest r (on the other split branch), BinaryOp add -> "output".
    for (const RawLayer& r : L) {
        static const char* ok[] = {"Input", "Split", "Convolution", "PReLU", "PixelShuffle", "Interp", "BinaryOp"};
        bool found = false;
        for (const char* o : ok) found |= (r.type == o);
        if (!found)
            return fail(err, "load_param: unsupported layer type '" + r.type + "' (" + r.name +
                                 "): only SRVGGNetCompact graphs are implemented on MI355X");
    }
    if (L[0].type != "Input" || L[0].out.size() != 1 || L[0].out[0] != "input")
        return fail(err, "load_param: first layer must be Input 'input'");
    if (L[1].type != "Split" || L[1].in.size() != 1 || L[1].in[0] != "input" || L[1].out.size() != 2)
        return fail(err, "load_param: expected Split of the input into two branches");
    const std::string br0 = L[1].out[0], br1 = L[1].out[1];
    size_t i = 2;
    std::string cur;
    bool expect_conv = true;
    for (; i < L.size(); ++i) {
        const RawLayer& r = L[i];
        if (r.type == "Convolution") {
            if (!expect_conv) return fail(err, "load_param: two Convolutions without PReLU at " + r.name);
            ConvWeights c;
            c.name = r.name;
            c.cout = r.geti(0, 0);
            c.weight_data_size = r.geti(6, 0);
            // ncnn convolution.cpp load_param: kernel_h (11), dilation_h (12), stride_h (13) default to the
            // _w values (1, 2, 3); pad_right (15) and pad_top (14) to pad_left (4), pad_bottom (16) to pad_top;
            // pad_value (18) to 0.  Anything but a symmetric 3x3 / dilation 1 / stride 1 / zero pad 1 layer
            // with a bias, no fused activation (9) and fp weights (8) is refused, not silently computed as one.
            const int pad_l = r.geti(4, 0), pad_t = r.geti(14, pad_l);
            if (r.geti(1, 0) != 3 || r.geti(11, 3) != 3 || pad_l != 1 || r.geti(5, 0) != 1 ||
                r.geti(2, 1) != 1 || r.geti(3, 1) != 1 || r.geti(9, 0) != 0 || r.geti(8, 0) != 0 ||
                r.geti(12, r.geti(2, 1)) != 1 || r.geti(13, r.geti(3, 1)) != 1 || pad_t != 1 ||
                r.geti(15, pad_l) != 1 || r.geti(16, pad_t) != 1 || r.getf(18, 0.0) != 0.0 || r.geti(7, 1) != 1)
                return fail(err, "load_param: " + r.name + " is not a plain 3x3/pad1/stride1/bias convolution");
            if (c.cout <= 0 || c.weight_data_size <= 0 || c.weight_data_size % (c.cout * 9))
                return fail(err, "load_param: bad sizes in " + r.name);
            c.cin = c.weight_data_size / (c.cout * 9);
            if (r.in.size() != 1 || r.out.size() != 1) return fail(err, "load_param: bad blobs in " + r.name);
            if (g.convs.empty()) {
                if (r.in[0] != br0 && r.in[0] != br1) return fail(err, "load_param: first conv must read a split branch");
                if (c.cin != 3) return fail(err, "load_param: first conv must have 3 input channels");
                g.nf = c.cout;
            } else {
                if (r.in[0] != cur) return fail(err, "load_param: broken chain at " + r.name);
                if (c.cin != g.nf) return fail(err, "load_param: channel mismatch at " + r.name);
            }
            cur = r.out[0];
            g.convs.push_back(c);
            expect_conv = false;
        } else if (r.type == "PReLU") {
            if (expect_conv || r.in.size() != 1 || r.in[0] != cur || r.out.size() != 1)
                return fail(err, "load_param: unexpected PReLU " + r.name);
            if (r.geti(0, 0) != g.convs.back().cout || g.convs.back().cout != g.nf)
                return fail(err, "load_param: PReLU slope count mismatch at " + r.name);
            g.prelu_names.push_back(r.name);
            cur = r.out[0];
            expect_conv = true;
        } else break;
    }
    if (g.convs.size() < 3 || expect_conv || g.prelu_names.size() != g.convs.size() - 1)
        return fail(err, "load_param: not a conv/PReLU stack ending in a convolution");
    if (i + 3 != L.size() || L[i].type != "PixelShuffle" || L[i + 1].type != "Interp" || L[i + 2].type != "BinaryOp")
        return fail(err, "load_param: expected PixelShuffle, Interp, BinaryOp tail");
    const RawLayer &ps = L[i], &ip = L[i + 1], &add = L[i + 2];
    const int r = ps.geti(0, 1);
    if (ps.geti(1, 0) != 0 || ps.in.size() != 1 || ps.in[0] != cur) return fail(err, "load_param: bad PixelShuffle");
    if (r != 1 && r != 2 && r != 4) return fail(err, "load_param: PixelShuffle factor must be 1, 2 or 4");
    if (g.convs.back().cout != 3 * r * r) return fail(err, "load_param: tail conv must have 3*r*r outputs");
    if (ip.geti(0, 0) != 1 || ip.getf(1, 1.0) != (double)r || ip.getf(2, 1.0) != (double)r ||
        ip.in.size() != 1 || (ip.in[0] != br0 && ip.in[0] != br1))
        return fail(err, "load_param: Interp must be nearest x r on the input branch");
    if (add.geti(0, 0) != 0 || add.in.size() != 2 || add.out.size() != 1 || add.out[0] != "output")
        return fail(err, "load_param: BinaryOp must be ADD producing 'output'");
    const bool a_ok = (add.in[0] == ps.out[0] && add.in[1] == ip.out[0]) || (add.in[1] == ps.out[0] && add.in[0] == ip.out[0]);
    if (!a_ok) return fail(err, "load_param: BinaryOp inputs must be PixelShuffle and Interp outputs");
    if (g.nf != 64 && g.nf != 24)
        return fail(err, "load_param: trunk width " + std::to_string(g.nf) + " has no MI355X kernel (64 and 24 do)");
    g.scale = r;
    g.param_loaded = true;
    return true;
}

bool load_bin(const std::string& path, Graph& g, std::string& err)
{
    if (!g.param_loaded) return fail(err, "load_model: load_param first");
    std::ifstream f(path, std::ios::binary);
    if (!f) return fail(err, "load_model: cannot open " + path);
    std::vector<char> raw((std::istreambuf_iterator<char>(f)), std::istreambuf_iterator<char>());
    g.bin_size = raw.size();
    size_t off = 0;
    auto need = [&](size_t n) { return off + n <= raw.size(); };
    g.slopes.clear();
    for (size_t ci = 0; ci < g.convs.size(); ++ci) {
        ConvWeights& c = g.convs[ci];
        const size_t n = (size_t)c.weight_data_size;
        if (!need(4)) return fail(err, "load_model: truncated at " + c.name);
        std::memcpy(&c.tag, raw.data() + off, 4);
        off += 4;
        c.w.resize(n);
        if (c.tag == 0x01306B47u) {
            const size_t bytes = (n * 2 + 3) & ~(size_t)3;
            if (!need(bytes)) return fail(err, "load_model: truncated at " + c.name);
            for (size_t k = 0; k < n; ++k) {
                uint16_t h;
                std::memcpy(&h, raw.data() + off + 2 * k, 2);
                c.w[k] = f16_bits_to_f32(h);
            }
            off += bytes;
        } else if (c.tag == 0) {
            if (!need(n * 4)) return fail(err, "load_model: truncated at " + c.name);
            std::memcpy(c.w.data(), raw.data() + off, n * 4);
            off += n * 4;
        } else {
            char b[64];
            std::snprintf(b, sizeof b, "0x%08X", c.tag);
            return fail(err, "load_model: unsupported weight flag " + std::string(b) + " at " + c.name);
        }
        c.bias.resize((size_t)c.cout);
        if (!need((size_t)c.cout * 4)) return fail(err, "load_model: truncated at " + c.name);
        std::memcpy(c.bias.data(), raw.data() + off, (size_t)c.cout * 4);
        off += (size_t)c.cout * 4;
        if (ci + 1 < g.convs.size()) {
            std::vector<float> s((size_t)g.nf);
            if (!need((size_t)g.nf * 4)) return fail(err, "load_model: truncated at " + g.prelu_names[ci]);
            std::memcpy(s.data(), raw.data() + off, (size_t)g.nf * 4);
            off += (size_t)g.nf * 4;
            g.slopes.push_back(s);
        }
    }
    g.bin_consumed = off;
    if (off != raw.size())
        return fail(err, "load_model: " + std::to_string(raw.size() - off) + " unread bytes in " + path +
                             " (weights do not match the graph)");
    g.model_loaded = true;
    return true;
}

void pack_conv3x3(const ConvWeights& c, int nf, std::vector<uint16_t>& out, int* ks_out, int* mf_out)
{
    const int spp = nf / 8, ko_n = 9 * spp, ks_n = (ko_n + 1) / 2, mf = (c.cout + 31) / 32;
    out.assign((size_t)ks_n * mf * 64 * 8, 0);
    for (int ks = 0; ks < ks_n; ++ks)
        for (int m = 0; m < mf; ++m)
            for (int lane = 0; lane < 64; ++lane) {
                const int i = lane & 31, half = lane >> 5;
                const int co = 32 * m + i, ko = 2 * ks + half;
                if (co >= c.cout || ko >= ko_n) continue;
                const int tap = ko / spp, oct = ko % spp;
                for (int e = 0; e < 8; ++e) {
                    const int ci = oct * 8 + e;
                    const float v = c.w[((size_t)co * c.cin + ci) * 9 + tap];
                    out[(((size_t)ks * mf + m) * 64 + lane) * 8 + e] = f32_to_f16_bits(v);
                }
            }
    if (ks_out) *ks_out = ks_n;
    if (mf_out) *mf_out = mf;
}

void pack_trunk64(const ConvWeights& c, std::vector<uint16_t>& out)
{
    out.assign((size_t)18 * 4 * 64 * 8, 0);
    for (int ks = 0; ks < 18; ++ks)
        for (int mb = 0; mb < 4; ++mb)
            for (int lane = 0; lane < 64; ++lane) {
                const int co = 16 * mb + (lane & 15), tap = ks >> 1;
                for (int e = 0; e < 8; ++e) {
                    const int ci = 32 * (ks & 1) + 8 * (lane >> 4) + e;
                    out[(((size_t)ks * 4 + mb) * 64 + lane) * 8 + e] = f32_to_f16_bits(c.w[((size_t)co * 64 + ci) * 9 + tap]);
                }
            }
}

void pack_trunk64_wino(const ConvWeights& c, std::vector<uint16_t>& out, const float* in_sign, const float* out_sign)
{
    out.assign((size_t)4 * 3 * 2 * 4 * 64 * 8, 0);
    for (int j = 0; j < 4; ++j)
        for (int dy = 0; dy < 3; ++dy)
            for (int ch = 0; ch < 2; ++ch)
                for (int mb = 0; mb < 4; ++mb)
                    for (int lane = 0; lane < 64; ++lane) {
                        const int co = 16 * mb + (lane & 15);
                        for (int e = 0; e < 8; ++e) {
                            const int ci = 32 * ch + 8 * (lane >> 4) + e;
                            const float* g = &c.w[((size_t)co * 64 + ci) * 9 + 3 * dy];
                            const double g0 = g[0], g1 = g[1], g2 = g[2];
                            const float u = j == 0 ? g[0] : j == 1 ? (float)(0.5 * (g0 + g1 + g2)) : j == 2 ? (float)(0.5 * (g0 - g1 + g2)) : g[2];
                            const float sg = (in_sign ? in_sign[ci] : 1.f) * (out_sign ? out_sign[co] : 1.f);       // (+-1: exact)
                            out[((((((size_t)j * 3 + dy) * 2 + ch) * 4 + mb) * 64) + lane) * 8 + e] = f32_to_f16_bits(sg * u);
                        }
                    }
}

void pack_tail64(const ConvWeights& c, std::vector<uint16_t>& out)
{
    const int mb_n = (c.cout + 15) / 16;
    out.assign((size_t)18 * mb_n * 64 * 8, 0);
    for (int ks = 0; ks < 18; ++ks)
        for (int mb = 0; mb < mb_n; ++mb)
            for (int lane = 0; lane < 64; ++lane) {
                const int co = 16 * mb + (lane & 15), tap = ks >> 1;
                if (co >= c.cout) continue;
                for (int e = 0; e < 8; ++e) {
                    const int ci = 32 * (ks & 1) + 8 * (lane >> 4) + e;
                    out[(((size_t)ks * mb_n + mb) * 64 + lane) * 8 + e] = f32_to_f16_bits(c.w[((size_t)co * 64 + ci) * 9 + tap]);
                }
            }
}

void pack_head(const ConvWeights& c, std::vector<uint16_t>& out, int* mf_out)
{
    const int mf = (c.cout + 31) / 32;
    out.assign((size_t)3 * mf * 64 * 8, 0);
    for (int ks = 0; ks < 3; ++ks)
        for (int m = 0; m < mf; ++m)
            for (int lane = 0; lane < 64; ++lane) {
                const int i = lane & 31, half = lane >> 5;
                const int co = 32 * m + i, o = 2 * ks + half;
                if (co >= c.cout) continue;
                for (int e = 0; e < 8; ++e) {
                    const int tap = 2 * o + (e >> 2), ch = e & 3;
                    if (tap >= 9 || ch >= 3) continue;
                    const float v = c.w[((size_t)co * 3 + ch) * 9 + tap];
                    out[(((size_t)ks * mf + m) * 64 + lane) * 8 + e] = f32_to_f16_bits(v);
                }
            }
    if (mf_out) *mf_out = mf;
}

void pack_sub16(const ConvWeights& c, std::vector<uint16_t>& out, int* ks_out, int* mb_out, const float* in_sign,
                const float* out_sign)
{
    const bool head = c.cin == 3;
    const int ks_n = head ? 2 : 7, mbn = (c.cout + 15) / 16;
    out.assign((size_t)ks_n * mbn * 64 * 8, 0);
    for (int ks = 0; ks < ks_n; ++ks)
        for (int mb = 0; mb < mbn; ++mb)
            for (int lane = 0; lane < 64; ++lane) {
                const int i = lane & 15, o = lane >> 4, ko = 4 * ks + o;
                // 24 output channels: block 1 holds channels 16..23 in rows 4q, 4q+1 (q = 0..3), so that every lane of
                // the MFMA result (rows 4q..4q+3) owns two real channels instead of four or none
                int co = 16 * mb + i;
                if (c.cout == 24 && mb == 1) co = (i & 3) < 2 ? 16 + 2 * (i >> 2) + (i & 3) : -1;
                // 3 output channels (the last layer): channel j in row 4j -- the first result register of lane group j --, so that
                // three lane groups finish one channel each and ONE byte store per pixel fragment writes 48 consecutive bytes
                if (c.cout == 3) co = (i & 3) == 0 && (i >> 2) < 3 ? (i >> 2) : -1;
                if (co < 0 || co >= c.cout) continue;
                for (int e = 0; e < 8; ++e) {
                    int tap, ci;
                    if (head) { tap = 2 * ko + (e >> 2); ci = e & 3; if (ci >= 3) continue; }
                    else { const int oct = SUB16_OCTET[ks][o]; tap = oct / 3; ci = 8 * (oct % 3) + e; }
                    if (tap >= 9) continue;
                    const float sg = (in_sign ? in_sign[ci] : 1.f) * (out_sign ? out_sign[co] : 1.f);   // +-1: exact
                    out[(((size_t)ks * mbn + mb) * 64 + lane) * 8 + e] = f32_to_f16_bits(sg * c.w[((size_t)co * c.cin + ci) * 9 + tap]);
                }
            }
    if (ks_out) *ks_out = ks_n;
    if (mb_out) *mb_out = mbn;
}

}  // namespace uva
