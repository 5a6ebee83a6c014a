// uva_generic.cpp -- parser / shape inference / weight reader for generic ncnn graphs (see uva_generic.h).
// File formats as in uva_model.cpp (ncnn src/net.cpp text .param, src/modelbin.cpp weight stream).
#include "uva_generic.h"

#include <cstdlib>
#include <cstring>
#include <algorithm>
#include <fstream>
#include <map>
#include <sstream>

namespace uva {
namespace {

bool gfail(std::string& err, const std::string& msg) { err = msg; return false; }

struct Raw {
    std::string type, name;
    std::vector<std::string> in, out;
    std::map<int, std::string> kv;
    int geti(int id, int def) const { auto it = kv.find(id); return it == kv.end() ? def : std::atoi(it->second.c_str()); }
    double getf(int id, double def) const { auto it = kv.find(id); return it == kv.end() ? def : std::atof(it->second.c_str()); }
    // array parameter "-233xx=n,v0,v1,..."
    std::vector<double> geta(int id) const
    {
        std::vector<double> v;
        auto it = kv.find(id);
        if (it == kv.end()) return v;
        std::stringstream ss(it->second);
        std::string tok;
        bool first = true;
        while (std::getline(ss, tok, ',')) {
            if (first) { first = false; continue; }   // the count
            v.push_back(std::atof(tok.c_str()));
        }
        return v;
    }
};

}  // namespace

bool parse_param_generic(const std::string& path, GenericGraph& g, std::string& err)
{
    g = GenericGraph();
    std::ifstream f(path);
    if (!f) return gfail(err, "load_param: cannot open " + path);
    long magic = 0;
    int nl = 0, nb = 0;
    f >> magic >> nl >> nb;
    if (!f || magic != 7767517) return gfail(err, "load_param: bad magic in " + path);
    if (nl < 2 || nl > 65536) return gfail(err, "load_param: bad layer count in " + path);
    std::string line;
    std::getline(f, line);
    std::vector<Raw> L;
    while ((int)L.size() < nl && std::getline(f, line)) {
        std::istringstream ss(line);
        Raw r;
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
    if ((int)L.size() != nl) return gfail(err, "load_param: truncated " + path);

    std::map<std::string, int> blob_id;
    auto blob_of = [&](const std::string& name, bool create) -> int {
        auto it = blob_id.find(name);
        if (it != blob_id.end()) return it->second;
        if (!create) return -1;
        GBlob b;
        b.name = name;
        g.blobs.push_back(b);
        blob_id[name] = (int)g.blobs.size() - 1;
        return (int)g.blobs.size() - 1;
    };
    auto root = [&](int b) { while (g.blobs[b].alias_of >= 0) b = g.blobs[b].alias_of; return b; };

    for (const Raw& r : L) {
        GLayer gl;
        gl.name = r.name;
        for (const auto& nm : r.in) {
            const int b = blob_of(nm, false);
            if (b < 0) return gfail(err, "load_param: " + r.name + " reads undefined blob " + nm);
            gl.in.push_back(b);
            g.blobs[root(b)].consumers += 1;
        }
        for (const auto& nm : r.out) {
            if (blob_id.count(nm)) return gfail(err, "load_param: blob " + nm + " defined twice");
            gl.out.push_back(blob_of(nm, true));
        }
        auto in_ch = [&](int k) { return g.blobs[gl.in[k]].channels; };
        auto in_sc = [&](int k) { return g.blobs[gl.in[k]].scale; };
        auto set_out = [&](int k, int ch, int sc) { g.blobs[gl.out[k]].channels = ch; g.blobs[gl.out[k]].scale = sc; };
        if (r.type == "Input") {
            if (r.in.size() != 0 || r.out.size() != 1 || g.in_blob >= 0) return gfail(err, "load_param: unexpected Input layer " + r.name);
            gl.kind = GLayer::INPUT;
            set_out(0, 3, 1);
            g.in_blob = gl.out[0];
        } else if (r.type == "Split") {
            if (r.in.size() != 1 || r.out.empty()) return gfail(err, "load_param: bad Split " + r.name);
            gl.kind = GLayer::SPLIT;
            g.blobs[root(gl.in[0])].consumers -= 1;      // a Split is not a consumer, its outputs' readers are
            for (size_t k = 0; k < gl.out.size(); ++k) {
                set_out((int)k, in_ch(0), in_sc(0));
                g.blobs[gl.out[k]].alias_of = root(gl.in[0]);
            }
        } else if (r.type == "Convolution") {
            if (r.in.size() != 1 || r.out.size() != 1) return gfail(err, "load_param: bad blobs in " + r.name);
            gl.kind = GLayer::CONV;
            ConvWeights c;
            c.name = r.name;
            c.cout = r.geti(0, 0);
            c.weight_data_size = r.geti(6, 0);
            const int kw = r.geti(1, 0), kh = r.geti(11, kw), pad = r.geti(4, 0);
            if (kw != kh || (kw != 1 && kw != 3)) return gfail(err, "load_param: " + r.name + ": only 1x1 and 3x3 kernels are implemented");
            if (pad != (kw == 3 ? 1 : 0) || r.geti(14, pad) != pad || r.geti(15, pad) != pad || r.geti(16, r.geti(14, pad)) != pad ||
                r.geti(2, 1) != 1 || r.geti(12, 1) != 1 || r.geti(3, 1) != 1 || r.geti(13, 1) != 1 || r.getf(18, 0.0) != 0.0 ||
                r.geti(8, 0) != 0 || r.geti(7, 1) != 1)
                return gfail(err, "load_param: " + r.name + " is not a stride-1 'same' convolution");
            gl.ksize = kw;
            gl.has_bias = r.geti(5, 0) != 0;
            const int act = r.geti(9, 0);
            if (act == 2) {
                const auto p = r.geta(-23310);
                gl.has_act = true;
                gl.act_slope = p.empty() ? 0.f : (float)p[0];
            } else if (act != 0) {
                return gfail(err, "load_param: " + r.name + ": fused activation type " + std::to_string(act) + " is not implemented");
            }
            c.cin = in_ch(0);
            if (c.cout <= 0 || c.cin <= 0 || c.weight_data_size != c.cout * c.cin * kw * kw)
                return gfail(err, "load_param: bad sizes in " + r.name);
            gl.conv = (int)g.convs.size();
            g.convs.push_back(c);
            set_out(0, c.cout, in_sc(0));
            g.flops_per_input_px += 2.0 * c.cout * c.cin * kw * kw * in_sc(0) * in_sc(0);
        } else if (r.type == "Concat") {
            if (r.in.size() < 2 || r.out.size() != 1 || r.geti(0, 0) != 0) return gfail(err, "load_param: bad Concat " + r.name);
            gl.kind = GLayer::CONCAT;
            int ch = 0;
            for (size_t k = 0; k < gl.in.size(); ++k) {
                if (in_sc((int)k) != in_sc(0)) return gfail(err, "load_param: Concat of different sizes at " + r.name);
                if (in_ch((int)k) % 8) return gfail(err, "load_param: Concat inputs must have a multiple of 8 channels at " + r.name);
                ch += in_ch((int)k);
            }
            set_out(0, ch, in_sc(0));
        } else if (r.type == "BinaryOp" || r.type == "Eltwise") {
            if (r.in.size() != 2 || r.out.size() != 1) return gfail(err, "load_param: bad blobs in " + r.name);
            if (in_ch(0) != in_ch(1) || in_sc(0) != in_sc(1)) return gfail(err, "load_param: shape mismatch at " + r.name);
            if (r.type == "BinaryOp") {
                if (r.geti(0, 0) != 0 || r.geti(1, 0) != 0) return gfail(err, "load_param: BinaryOp " + r.name + " is not a tensor ADD");
                gl.kind = GLayer::ADD;
                gl.coeffs = {1.f, 1.f};
            } else {
                if (r.geti(0, 0) != 1) return gfail(err, "load_param: Eltwise " + r.name + " is not SUM");
                gl.kind = GLayer::ELTWISE_SUM;
                const auto p = r.geta(-23301);
                if (p.empty()) gl.coeffs = {1.f, 1.f};
                else if (p.size() == 2) gl.coeffs = {(float)p[0], (float)p[1]};
                else return gfail(err, "load_param: Eltwise " + r.name + " needs two coefficients");
            }
            set_out(0, in_ch(0), in_sc(0));
        } else if (r.type == "Interp") {
            if (r.in.size() != 1 || r.out.size() != 1) return gfail(err, "load_param: bad blobs in " + r.name);
            const double sh = r.getf(1, 1.0), sw = r.getf(2, 1.0);
            if (r.geti(0, 0) != 1 || sh != sw || sh != (double)(int)sh || sh < 1 || sh > 8)
                return gfail(err, "load_param: Interp " + r.name + " must be nearest with an integer scale");
            gl.kind = GLayer::INTERP_NEAREST;
            gl.factor = (int)sh;
            set_out(0, in_ch(0), in_sc(0) * gl.factor);
        } else if (r.type == "PReLU") {
            if (r.in.size() != 1 || r.out.size() != 1 || r.geti(0, 0) != in_ch(0)) return gfail(err, "load_param: bad PReLU " + r.name);
            gl.kind = GLayer::PRELU;
            gl.slopes = (int)g.prelu_sizes.size();
            g.prelu_sizes.push_back(in_ch(0));
            set_out(0, in_ch(0), in_sc(0));
        } else if (r.type == "PixelShuffle") {
            const int f = r.geti(0, 1);
            if (r.in.size() != 1 || r.out.size() != 1 || r.geti(1, 0) != 0 || f < 1 || f > 8 || in_ch(0) % (f * f))
                return gfail(err, "load_param: bad PixelShuffle " + r.name);
            gl.kind = GLayer::PIXELSHUFFLE;
            gl.factor = f;
            set_out(0, in_ch(0) / (f * f), in_sc(0) * f);
        } else {
            return gfail(err, "load_param: unsupported layer type '" + r.type + "' (" + r.name + ")");
        }
        g.layers.push_back(gl);
    }
    if (g.in_blob < 0 || g.blobs[g.in_blob].name != "input") return gfail(err, "load_param: no Input layer named 'input'");
    g.out_blob = blob_of("output", false);
    if (g.out_blob < 0) return gfail(err, "load_param: no blob named 'output'");
    if (g.blobs[g.out_blob].channels != 3) return gfail(err, "load_param: 'output' must have 3 channels");
    g.blobs[root(g.out_blob)].consumers += 1;            // the caller reads it
    g.scale = g.blobs[g.out_blob].scale;
    for (const auto& b : g.blobs) g.max_channels = b.channels > g.max_channels ? b.channels : g.max_channels;
    plan_concat_groups(g);
    g.param_loaded = true;
    return true;
}

void plan_concat_groups(GenericGraph& g)
{
    const int nb = (int)g.blobs.size(), nl = (int)g.layers.size();
    auto root = [&](int b) { while (g.blobs[b].alias_of >= 0) b = g.blobs[b].alias_of; return b; };
    std::vector<int> producer(nb, -1);
    std::vector<std::vector<int>> readers(nb);
    for (int li = 0; li < nl; ++li) {
        const GLayer& gl = g.layers[li];
        if (gl.kind == GLayer::SPLIT) continue;
        for (int b : gl.out) producer[b] = li;
        for (int b : gl.in) readers[root(b)].push_back(li);
    }
    // a convolution g_conv3_lds takes (csrc/uva_generic.hip.h): 3x3, <= 64 output channels, <= 192 input channels
    auto lds_conv = [&](int li, int cin) {
        const GLayer& gl = g.layers[li];
        return gl.kind == GLayer::CONV && gl.ksize == 3 && g.blobs[gl.out[0]].channels <= 64 && cin <= 192 && cin % 32 == 0;
    };
    std::vector<char> used(nl, 0);
    for (int l0 = 0; l0 < nl; ++l0) {
        if (g.layers[l0].kind != GLayer::CONCAT || used[l0] || g.layers[l0].in.size() != 2) continue;
        std::vector<int> chain{l0}, cur;
        for (int b : g.layers[l0].in) cur.push_back(root(b));
        for (;;) {      // the next Concat that extends `cur` by one blob
            int next = -1;
            for (int lj = chain.back() + 1; lj < nl && next < 0; ++lj) {
                const GLayer& c = g.layers[lj];
                if (c.kind != GLayer::CONCAT || used[lj] || c.in.size() != cur.size() + 1) continue;
                bool same = true;
                for (size_t k = 0; k < cur.size() && same; ++k) same = root(c.in[k]) == cur[k];
                if (same) next = lj;
            }
            if (next < 0) break;
            chain.push_back(next);
            cur.push_back(root(g.layers[next].in.back()));
        }
        // validation.  A member is written by a convolution g_conv3_lds takes or by an element-wise sum (both can write
        // a channel range of a wider array) and read by the chain's Concats, element-wise sums, or such convolutions.
        bool ok = g.blobs[cur[0]].channels % 8 == 0;
        const int scale = g.blobs[cur[0]].scale;
        for (size_t k = 1; k < cur.size() && ok; ++k) {
            const int y = cur[k], pl = producer[y];
            ok = pl >= 0 && g.blobs[y].alias_of < 0 && g.blobs[y].group < 0 && g.blobs[y].scale == scale && g.blobs[y].channels % 8 == 0;
            if (!ok) break;
            const GLayer& pr = g.layers[pl];
            ok = pr.kind == GLayer::ADD || pr.kind == GLayer::ELTWISE_SUM || lds_conv(pl, g.blobs[root(pr.in[0])].channels);
            int in_chain = 0;
            for (int r : readers[y]) {
                const GLayer& rd = g.layers[r];
                if (std::find(chain.begin(), chain.end(), r) != chain.end()) ++in_chain;
                else ok = ok && (rd.kind == GLayer::ADD || rd.kind == GLayer::ELTWISE_SUM || lds_conv(r, g.blobs[y].channels));
            }
            // a member feeds every Concat of the chain from its own on, once each
            ok = ok && in_chain == (int)chain.size() - (int)(k - 1);
        }
        // The chain's first blob x (a dense block's input) joins too -- the first Concat then copies nothing -- if it is
        // written by a sum or a convolution of g_conv3_lds and all its other readers take a channel range of a wider
        // array: sums, and 3x3 / 1x1 convolutions of that kernel.
        bool x_joins = ok;
        if (x_joins) {
            const int x = cur[0], pl = producer[x];
            auto conv_any = [&](int li, int cin) {
                const GLayer& gl = g.layers[li];
                return gl.kind == GLayer::CONV && g.blobs[gl.out[0]].channels <= 64 && cin <= 192 && cin % 32 == 0;
            };
            x_joins = pl >= 0 && g.blobs[x].alias_of < 0 && g.blobs[x].group < 0 && x != g.out_blob;
            if (x_joins) {
                const GLayer& pr = g.layers[pl];
                // (a producing convolution only has to be one that can WRITE a channel range: what it reads is its own
                // business -- the 3-channel head convolution in front of 4x_Valar_v1's first dense block, padded to 32)
                x_joins = pr.kind == GLayer::ADD || pr.kind == GLayer::ELTWISE_SUM ||
                          conv_any(pl, (g.blobs[root(pr.in[0])].channels + 31) / 32 * 32);
            }
            int in_chain = 0;
            for (int r : readers[x]) {
                const GLayer& rd = g.layers[r];
                if (std::find(chain.begin(), chain.end(), r) != chain.end()) ++in_chain;
                else x_joins = x_joins && (rd.kind == GLayer::ADD || rd.kind == GLayer::ELTWISE_SUM || conv_any(r, g.blobs[x].channels));
            }
            x_joins = x_joins && in_chain == (int)chain.size();
        }
        for (size_t j = 0; j < chain.size() && ok; ++j) {
            const int o = g.layers[chain[j]].out[0];
            ok = g.blobs[o].alias_of < 0 && g.blobs[o].channels % 32 == 0 && g.blobs[o].channels <= 192 && o != g.out_blob;
            for (int r : readers[o]) ok = ok && lds_conv(r, g.blobs[o].channels);
        }
        if (!ok) continue;
        const int gid = (int)g.group_channels.size();
        g.group_channels.push_back(g.blobs[g.layers[chain.back()].out[0]].channels);
        int off = g.blobs[cur[0]].channels, count = 0;
        if (x_joins) {
            g.blobs[cur[0]].group = gid;
            g.blobs[cur[0]].group_off = 0;
            ++count;
        }
        for (size_t k = 1; k < cur.size(); ++k) {
            g.blobs[cur[k]].group = gid;
            g.blobs[cur[k]].group_off = off;
            off += g.blobs[cur[k]].channels;
            ++count;
        }
        for (size_t j = 0; j < chain.size(); ++j) {
            GLayer& c = g.layers[chain[j]];
            c.concat_mode = (j == 0 && !x_joins) ? 1 : 2;
            g.blobs[c.out[0]].group = gid;
            g.blobs[c.out[0]].group_off = 0;
            used[chain[j]] = 1;
            ++count;
        }
        g.group_blobs.push_back(count);
    }
}

std::vector<RdbMatch> find_rdbs(const GenericGraph& g)
{
    std::vector<RdbMatch> out;
    const int nb = (int)g.blobs.size(), nl = (int)g.layers.size();
    auto root = [&](int b) { while (g.blobs[b].alias_of >= 0) b = g.blobs[b].alias_of; return b; };
    std::vector<int> producer(nb, -1);
    for (int li = 0; li < nl; ++li)
        if (g.layers[li].kind != GLayer::SPLIT)
            for (int b : g.layers[li].out) producer[b] = li;
    for (int gi = 0; gi < (int)g.group_channels.size(); ++gi) {
        if (g.group_channels[gi] != 192) continue;
        // the chain's members by channel offset (Concat outputs sit at offset 0 with more than 64 channels)
        int x = -1, xk[4] = {-1, -1, -1, -1}, cat[5] = {-1, -1, -1, -1, -1};      // cat[k]: the Concat blob with 64 + 32k channels
        for (int b = 0; b < nb; ++b) {
            const GBlob& bl = g.blobs[b];
            if (bl.group != gi || bl.alias_of >= 0) continue;
            if (bl.group_off == 0 && bl.channels == 64) x = b;
            else if (bl.group_off == 0 && bl.channels > 64 && (bl.channels - 64) % 32 == 0 && bl.channels <= 192) cat[(bl.channels - 64) / 32] = b;
            else if (bl.channels == 32 && bl.group_off >= 64 && (bl.group_off - 64) % 32 == 0) xk[(bl.group_off - 64) / 32] = b;
        }
        if (x < 0 || xk[0] < 0 || xk[1] < 0 || xk[2] < 0 || xk[3] < 0 || cat[1] < 0 || cat[2] < 0 || cat[3] < 0) continue;
        RdbMatch m;
        m.group = gi;
        auto conv3_of = [&](int li, int in_blob, float* slope) {      // a 3x3 convolution with bias and LeakyReLU reading in_blob, 32 outputs
            if (li < 0) return false;
            const GLayer& l = g.layers[li];
            if (l.kind != GLayer::CONV || l.ksize != 3 || !l.has_bias || !l.has_act || root(l.in[0]) != in_blob) return false;
            if (g.convs[l.conv].cout != 32) return false;
            *slope = l.act_slope;
            return true;
        };
        float s1 = 0, s2 = 0, s3 = 0, s4 = 0;
        m.c1 = producer[xk[0]];
        if (!conv3_of(m.c1, x, &s1)) continue;
        m.add2 = producer[xk[1]];
        m.c3 = producer[xk[2]];
        m.add4 = producer[xk[3]];
        if (m.add2 < 0 || m.add4 < 0 || !conv3_of(m.c3, cat[2], &s3)) continue;
        const GLayer& a2 = g.layers[m.add2];
        const GLayer& a4 = g.layers[m.add4];
        auto plain_add = [](const GLayer& l) {
            return (l.kind == GLayer::ADD || l.kind == GLayer::ELTWISE_SUM) && l.in.size() == 2 && l.coeffs.size() == 2 && l.coeffs[0] == 1.f && l.coeffs[1] == 1.f;
        };
        if (!plain_add(a2) || !plain_add(a4)) continue;
        // x2 = conv3(cat(x, x1)) [first operand] + conv1(x) [second]; x4 = conv3(cat(x..x3)) + x2
        const int pa = root(a2.in[0]), pb = root(a2.in[1]), pc = root(a4.in[0]);
        if (g.blobs[pa].consumers != 1 || g.blobs[pb].consumers != 1 || g.blobs[pc].consumers != 1 || root(a4.in[1]) != xk[1]) continue;
        if (g.blobs[pa].group >= 0 || g.blobs[pb].group >= 0 || g.blobs[pc].group >= 0) continue;
        m.c2 = producer[pa];
        m.c2s = producer[pb];
        m.c4 = producer[pc];
        if (!conv3_of(m.c2, cat[1], &s2) || !conv3_of(m.c4, cat[3], &s4)) continue;
        if (m.c2s < 0) continue;
        const GLayer& ls = g.layers[m.c2s];
        if (ls.kind != GLayer::CONV || ls.ksize != 1 || ls.has_bias || ls.has_act || root(ls.in[0]) != x || g.convs[ls.conv].cout != 32) continue;
        if (s1 != s2 || s1 != s3 || s1 != s4) continue;
        if (!(s1 >= 0.f && s1 <= 1.f)) continue;     // rdb4_kernel's LeakyReLU is max(v, slope * v): the layer-by-layer kernels take the others
        // program order: everything the launch at c1 replaces comes after it, and x is written before it
        if (!(m.c1 < m.c2 && m.c1 < m.c2s && m.c2 < m.add2 && m.c2s < m.add2 && m.add2 < m.c3 && m.c3 < m.c4 && m.c4 < m.add4)) continue;
        if (producer[x] < 0 || producer[x] >= m.c1) continue;
        m.slope = s1;
        out.push_back(m);
    }
    return out;
}

bool load_bin_generic(const std::string& path, GenericGraph& g, std::string& err)
{
    if (!g.param_loaded) return gfail(err, "load_model: load_param first");
    std::ifstream f(path, std::ios::binary);
    if (!f) return gfail(err, "load_model: cannot open " + path);
    std::vector<char> raw((std::istreambuf_iterator<char>(f)), std::istreambuf_iterator<char>());
    size_t off = 0;
    auto need = [&](size_t n) { return off + n <= raw.size(); };
    g.prelu.assign(g.prelu_sizes.size(), {});
    for (const GLayer& gl : g.layers) {
        if (gl.kind == GLayer::CONV) {
            ConvWeights& c = g.convs[gl.conv];
            const size_t n = (size_t)c.weight_data_size;
            if (!need(4)) return gfail(err, "load_model: truncated at " + c.name);
            std::memcpy(&c.tag, raw.data() + off, 4);
            off += 4;
            c.w.resize(n);
            if (c.tag == 0x01306B47u) {
                const size_t bytes = (n * 2 + 3) & ~(size_t)3;
                if (!need(bytes)) return gfail(err, "load_model: truncated at " + c.name);
                for (size_t k = 0; k < n; ++k) {
                    uint16_t h;
                    std::memcpy(&h, raw.data() + off + 2 * k, 2);
                    c.w[k] = f16_bits_to_f32(h);
                }
                off += bytes;
            } else if (c.tag == 0) {
                if (!need(n * 4)) return gfail(err, "load_model: truncated at " + c.name);
                std::memcpy(c.w.data(), raw.data() + off, n * 4);
                off += n * 4;
            } else {
                return gfail(err, "load_model: unsupported weight flag at " + c.name);
            }
            c.bias.assign((size_t)c.cout, 0.f);
            if (gl.has_bias) {
                if (!need((size_t)c.cout * 4)) return gfail(err, "load_model: truncated at " + c.name);
                std::memcpy(c.bias.data(), raw.data() + off, (size_t)c.cout * 4);
                off += (size_t)c.cout * 4;
            }
        } else if (gl.kind == GLayer::PRELU) {
            const size_t n = (size_t)g.prelu_sizes[gl.slopes];
            if (!need(n * 4)) return gfail(err, "load_model: truncated at " + gl.name);
            g.prelu[gl.slopes].resize(n);
            std::memcpy(g.prelu[gl.slopes].data(), raw.data() + off, n * 4);
            off += n * 4;
        }
    }
    if (off != raw.size())
        return gfail(err, "load_model: " + std::to_string(raw.size() - off) + " unread bytes in " + path +
                              " (weights do not match the graph)");
    g.model_loaded = true;
    return true;
}

// 3x3 weights as 1-D Winograd F(2,3) along x for g_conv3_sww (csrc/uva_sww.hip.h): per filter row dy the three taps g0, g1, g2
// become U0 = g0, U1 = (g0 + g1 + g2) / 2, U2 = (g0 - g1 + g2) / 2, U3 = g2 (fp32 arithmetic on the stored weights, ONE rounding
// to fp16 each); image [dy*4 + j][cin_pad/32][cout_pad/16][64 lanes][8], natural octet order like pack_generic's.
void pack_generic_wino(const ConvWeights& c, int cin_pad, int cout_pad, std::vector<uint16_t>& out)
{
    const int c32n = cin_pad / 32, mbn = cout_pad / 16;
    out.assign((size_t)12 * c32n * mbn * 64 * 8, 0);
    for (int dy = 0; dy < 3; ++dy)
        for (int c32 = 0; c32 < c32n; ++c32)
            for (int mb = 0; mb < mbn; ++mb)
                for (int lane = 0; lane < 64; ++lane) {
                    const int co = 16 * mb + (lane & 15);
                    if (co >= c.cout) continue;
                    for (int e = 0; e < 8; ++e) {
                        const int ci = 32 * c32 + 8 * (lane >> 4) + e;
                        if (ci >= c.cin) continue;
                        const float* g = &c.w[((size_t)co * c.cin + ci) * 9 + dy * 3];
                        // (the stored weights are fp16 values or fp32 ones: either way the transform is taken in fp32)
                        const float u[4] = {g[0], (g[0] + g[1] + g[2]) * 0.5f, (g[0] - g[1] + g[2]) * 0.5f, g[2]};
                        for (int j = 0; j < 4; ++j)
                            out[((((size_t)(dy * 4 + j) * c32n + c32) * mbn + mb) * 64 + lane) * 8 + e] = f32_to_f16_bits(u[j]);
                    }
                }
}

void pack_generic(const ConvWeights& c, int ksize, int cin_pad, int cout_pad, std::vector<uint16_t>& out, bool lds_order)
{
    const int taps = ksize * ksize, c32n = cin_pad / 32, mbn = cout_pad / 16;
    static const int unit_of_group[4] = {0, 2, 1, 3};       // g_conv3_lds: K-octet group o reads the 16-byte unit {0,2,1,3}[o]
    out.assign((size_t)taps * c32n * mbn * 64 * 8, 0);
    for (int tap = 0; tap < taps; ++tap)
        for (int c32 = 0; c32 < c32n; ++c32)
            for (int mb = 0; mb < mbn; ++mb)
                for (int lane = 0; lane < 64; ++lane) {
                    const int co = 16 * mb + (lane & 15);
                    if (co >= c.cout) continue;
                    const int oct = lds_order ? unit_of_group[lane >> 4] : (lane >> 4);
                    for (int e = 0; e < 8; ++e) {
                        const int ci = 32 * c32 + 8 * oct + e;
                        if (ci >= c.cin) continue;
                        out[((((size_t)tap * c32n + c32) * mbn + mb) * 64 + lane) * 8 + e] =
                            f32_to_f16_bits(c.w[((size_t)co * c.cin + ci) * taps + tap]);
                    }
                }
}

}  // namespace uva
