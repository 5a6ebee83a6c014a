// uva_api.hip -- C ABI (include/uva.h) and device runtime of libuva.so.
//
// Replaces the ncnn_vulkan surface used by the reference worker functions
// (upscale/upscale_processing.py:54-73 init_worker, :258-299 apply_model, :395-477 process_tile,
// :480-542 upscale_image; test_gpus.py:47-67 enumeration).  One uva_net = one ncnn.Net: it owns a
// HIP stream, the packed weights, and a small cache of per-geometry workspaces (zero-bordered
// fp16 NHWC ping-pong planes) sized for 288 GB of HBM: nothing is freed between frames.
#include "../../include/uva.h"
#include "uva_denoise.hip.h"
#include "uva_generic.hip.h"
#include "uva_kernels.hip.h"
#include "uva_model.h"
#include "uva_png.hip.h"
#include "uva_wino.h"
#include "uva_sub5.h"
#include "uva_sub10.h"

namespace uva {   // uva_pngread.cpp
int png_read_bgr(const uint8_t* file, size_t len, uint8_t* out, size_t cap, int* h_out, int* w_out, std::string& err);
int zlib_decompress_exact(const uint8_t* in, size_t n, uint8_t* out, size_t out_len, std::string& err);
}

#include <hip/hip_runtime.h>

#include <algorithm>
#include <climits>
#include <cmath>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <list>
#include <memory>
#include <mutex>
#include <set>
#include <string>
#include <vector>

using namespace uva;

namespace {

thread_local std::string g_err;

int fail(const std::string& msg)
{
    g_err = msg;
    return 1;
}

#define HIP_TRY(expr)                                                                          \
    do {                                                                                       \
        hipError_t e_ = (expr);                                                                \
        if (e_ != hipSuccess)                                                                  \
            return fail(std::string(#expr) + ": " + hipGetErrorString(e_));                    \
    } while (0)

struct Workspace {
    int h = 0, w = 0, tile_size = 0, border = 0;
    std::vector<PlaneDesc> planes;
    PlaneDesc* d_planes = nullptr;
    uint4* d_sched4 = nullptr;   // per 4-row work tile: trunk_kernel's schedule entry (ConvArgs::sched4)
    int ntiles = 0;     // 8-row work tiles (head, tail, 24-feature trunk)
    int ntiles4 = 0;    // 4-row work tiles (64-feature trunk kernel)
    size_t act_pixels = 0;
    // Activation buffers: act_base[i] is the allocation, act[i] = act_base[i] + guard the planes.  The guard
    // (zeroed, never written) on both ends keeps trunk2_kernel's halo reads of rows -2 / h+4 and columns
    // -2 / pitch+1 -- which only ever feed pixels it masks to zero -- inside the allocation.
    size_t guard_bytes = 0;
    size_t alloc_bytes = 0;      // device bytes of the two activation buffers (the cache's LRU budget)
    char* act_base[2] = {nullptr, nullptr};
    _Float16* act[2] = {nullptr, nullptr};
    // sub10_kernel (the whole 24-feature 1x net in one launch): per-workgroup row descriptors
    // (one set per number of frames in a launch, built on first use: [k - 1] = k frames; uva_net_process_u8_device_batch)
    uint4* d_rows10[S10_MAXB] = {};
    int* d_nrows10[S10_MAXB] = {};
    int max_rows10[S10_MAXB] = {}, grid10 = 0;
    bool sub10_unfit = false;             // one frame does not fit the kernel's row table: the per-pair path
    int sub10_max_batch = S10_MAXB;       // the largest number of frames whose row table fits (found on demand)
    // sub5_kernel (the 1x net as two launches of five layers, two pipelines per workgroup): row descriptors and the 24-channel
    // image between the launches
    uint4* d_rows5 = nullptr;
    int* d_nrows5 = nullptr;
    int max_rows5 = 0, grid5 = 0;
    char* d_mid5 = nullptr;
    bool sub5_unfit = false;
    // trunk2_kernel (fused layer pair): per-workgroup step lists
    Trunk2Step* d_steps2 = nullptr;
    int* d_nsteps2 = nullptr;
    int max_steps2 = 0, grid2 = 0;
    // trunkw_kernel (fused layer pair, Winograd F(2,3)): its own step lists (segments start without the shared rows)
    Trunk2Step* d_stepsw = nullptr;
    int* d_nstepsw = nullptr;
    int max_stepsw = 0;
    void release()
    {
        if (d_planes) (void)hipFree(d_planes);
        if (d_sched4) (void)hipFree(d_sched4);
        if (d_steps2) (void)hipFree(d_steps2);
        if (d_nsteps2) (void)hipFree(d_nsteps2);
        if (d_stepsw) (void)hipFree(d_stepsw);
        if (d_nstepsw) (void)hipFree(d_nstepsw);
        d_stepsw = nullptr;
        d_nstepsw = nullptr;
        for (auto& q : d_rows10) { if (q) (void)hipFree(q); q = nullptr; }
        for (auto& q : d_nrows10) { if (q) (void)hipFree(q); q = nullptr; }
        if (d_rows5) (void)hipFree(d_rows5);
        if (d_nrows5) (void)hipFree(d_nrows5);
        if (d_mid5) (void)hipFree(d_mid5);
        d_rows5 = nullptr;
        d_nrows5 = nullptr;
        d_mid5 = nullptr;
        d_sched4 = nullptr;
        d_steps2 = nullptr;
        d_nsteps2 = nullptr;
        if (act_base[0]) (void)hipFree(act_base[0]);
        if (act_base[1]) (void)hipFree(act_base[1]);
        d_planes = nullptr;
        act_base[0] = act_base[1] = nullptr;
        act[0] = act[1] = nullptr;
    }
};

struct DeviceLayer {
    half8* wpk = nullptr;
    half8* wpk_w = nullptr;   // 64 -> 64 trunk layers: pack_trunk64_wino image for trunkw_kernel
    half8* wpk_wn = nullptr;  // ... the same for a launch's FIRST layer whose input arrives with the previous layer's slope > 1 channels
                              // still negated (the launch before it did not restore their sign: run_graph, `carry`)
    float* bias_w = nullptr;  // ... and its bias, negated for the channels trunkw_kernel computes negated (uva_wino.h TW_ACT_F16)
    bool flip_w = false;      // ... which this layer has (a PReLU slope above 1)
    half8* wpk16 = nullptr;   // tail layer of the 64-feature 2x / 4x nets: pack_tail64 image for tail_kernel / tail4_kernel
    half8* wpk_s10 = nullptr; // 24-feature 1x net: pack_sub16 image for sub10_kernel, with its own (sign-folded) bias
    float* bias_s10 = nullptr;
    float* bias = nullptr;
    float* slope = nullptr;
};

struct LastCall {
    bool valid = false;
    Workspace* ws = nullptr;
    bool f32 = false;
    const void* src = nullptr;
    size_t src_stride = 0;
    void* dst = nullptr;       // u8 route: device result buffer of the call
    size_t dst_stride = 0;
};

std::mutex g_nets_mu;
std::set<uva_net*> g_nets;

}  // namespace

struct uva_net {
    int device = 0;
    Graph g;
    // graphs outside the SRVGGNetCompact pattern (4x_Valar_v1, `-m r`): generic layer-by-layer executor
    bool generic = false;
    GenericGraph gg;
    GenericDevice gd;
    bool dev_ready = false;
    bool dev_partial = false;     // some device state exists (ensure_device started); free_device() must run
    int ncu = 256;
    hipStream_t stream = nullptr;
    std::vector<DeviceLayer> layers;
    std::list<Workspace> wss;   // most recently used first
    // staging for the f32 (Extractor) entry point
    float *d_fin = nullptr, *d_fout = nullptr;
    size_t d_fin_cap = 0, d_fout_cap = 0;
    _Float16* d_sink = nullptr;   // where out-of-image lanes of the trunk kernel store to
    bool attr_set[40] = {false};  // hipFuncAttributeMaxDynamicSharedMemorySize done for kernel slot k on this net's device
    int last_act_buf = 0;         // which ping-pong buffer the last run_graph() left its last trunk activation in
    bool generic_fuse_add = true; // generic graphs: sums that follow a convolution are done in its epilogue (UVA_GENERIC_FUSE_ADD=0: own launch)
    bool generic_lds_conv = true; // generic graphs: 3x3 convolutions through g_conv3_lds (UVA_GENERIC_LDS=0: the plain g_conv<3>)
    bool fuse_all = true;         // 24-feature 1x net: all ten convolutions in one launch (sub10_kernel); UVA_SUB10=0 turns it off
    bool split5 = false;          // ... as two launches of five layers instead (sub5_kernel, csrc/uva_sub5.hip.h); UVA_SUB5=1
    bool fuse_pairs = true;       // 64-feature nets: trunk layers run two per launch (trunk2_kernel); UVA_TRUNK_FUSION=0 turns it off
    bool carry = true;            // ... and between two launches the negated channels stay negated in HBM (UVA_TW_CARRY=0: every launch
                                  // restores the signs in front of its stores)
    bool act16 = true;            // trunkw_kernel: PReLU on packed fp16 (uva_wino.h TW_ACT_F16); UVA_TW_ACT16=0: on the fp32 sums
    bool wino = true;             // ... as 1-D Winograd F(2,3) (trunkw_kernel); UVA_TRUNK_WINO=0: trunk2_kernel (direct convolution)
    LastCall last;
    // pipelined host route (uva_net_submit_u8 / uva_net_collect_u8): H2D, kernels and D2H of
    // consecutive frames overlap on three streams; PIPE_SLOTS frames may be in flight
    static constexpr int PIPE_SLOTS = 3;
    struct PipeSlot {
        bool busy = false;
        long long ticket = -1;
        uint8_t *d_in = nullptr, *d_out = nullptr, *h_in = nullptr, *h_out = nullptr;   // h_*: pinned staging
        uint8_t* d_png = nullptr;        // the PNG encoder's blocks: [meta][slots], then [the blocks packed end to end]
        size_t d_in_cap = 0, d_out_cap = 0, h_in_cap = 0, h_out_cap = 0, d_png_cap = 0;
        uint8_t* png_ws = nullptr;       // PNG submit: the caller's workspace, how many packed bytes went there with the
        size_t png_sent = 0;             // frame's download, and the frame size (collect fetches the rest, if any)
        int png_h = 0, png_w = 0;
        hipEvent_t ev_h2d = nullptr, ev_done = nullptr, ev_d2h = nullptr;
        static constexpr int BANDS = 4;  // a staged (pageable) result comes down in row bands: collect copies band k to the
        hipEvent_t ev_band[BANDS] = {nullptr, nullptr, nullptr, nullptr};   // caller while band k + 1 is still on PCIe
        uint8_t* user_out = nullptr;     // where collect copies the staged result (null: D2H went there directly)
        size_t user_out_stride = 0, out_row = 0;
        int out_rows = 0;
    };
    PipeSlot pipe[PIPE_SLOTS];
    size_t png_guess = 0;                // packed bytes of the last PNG frame collected: the next download's size
    hipStream_t s_h2d = nullptr, s_d2h = nullptr;
    long long next_ticket = 0;
    // profiling
    bool prof = false;
    std::vector<hipEvent_t> ev_free;        // timing-only events (no system-scope fence: take_event)
    std::vector<hipEvent_t> ev_sync_free;   // ordering events between streams (take_sync_event)
    struct EvSet { hipEvent_t e[4]; int ntrunk; };
    std::vector<EvSet> ev_pending;
    struct EvPair { hipEvent_t a, b; int kind; };       // generic graphs: one launch of rdb4_kernel (kind 1) / the 192 -> 64 convolution (2)
    std::vector<EvPair> ev_pairs;
    long long launches[3] = {0, 0, 0};
    double total_ms[3] = {0, 0, 0};

    void free_device()
    {
        if (!dev_ready && !dev_partial) return;
        dev_partial = false;
        std::memset(attr_set, 0, sizeof attr_set);
        (void)hipSetDevice(device);
        if (stream) (void)hipStreamSynchronize(stream);
        for (auto& l : layers) {
            if (l.wpk) (void)hipFree(l.wpk);
            if (l.wpk_w) (void)hipFree(l.wpk_w);
            if (l.bias_w) (void)hipFree(l.bias_w);
            if (l.wpk_wn) (void)hipFree(l.wpk_wn);
            if (l.wpk16) (void)hipFree(l.wpk16);
            if (l.wpk_s10) (void)hipFree(l.wpk_s10);
            if (l.bias_s10) (void)hipFree(l.bias_s10);
            if (l.bias) (void)hipFree(l.bias);
            if (l.slope) (void)hipFree(l.slope);
        }
        layers.clear();
        for (auto& w : wss) w.release();
        wss.clear();
        gd.release();
        if (d_fin) (void)hipFree(d_fin);
        if (d_fout) (void)hipFree(d_fout);
        if (d_sink) (void)hipFree(d_sink);
        d_sink = nullptr;
        if (s_h2d) (void)hipStreamSynchronize(s_h2d);
        if (s_d2h) (void)hipStreamSynchronize(s_d2h);
        for (auto& ps : pipe) {
            if (ps.d_in) (void)hipFree(ps.d_in);
            if (ps.d_out) (void)hipFree(ps.d_out);
            if (ps.d_png) (void)hipFree(ps.d_png);
            if (ps.h_in) (void)hipHostFree(ps.h_in);
            if (ps.h_out) (void)hipHostFree(ps.h_out);
            if (ps.ev_h2d) (void)hipEventDestroy(ps.ev_h2d);
            if (ps.ev_done) (void)hipEventDestroy(ps.ev_done);
            if (ps.ev_d2h) (void)hipEventDestroy(ps.ev_d2h);
            for (auto& e : ps.ev_band)
                if (e) (void)hipEventDestroy(e);
            ps = PipeSlot();
        }
        if (s_h2d) (void)hipStreamDestroy(s_h2d);
        if (s_d2h) (void)hipStreamDestroy(s_d2h);
        s_h2d = s_d2h = nullptr;
        d_fin = d_fout = nullptr;
        d_fin_cap = d_fout_cap = 0;
        for (auto& s : ev_pending)
            for (auto e : s.e) (void)hipEventDestroy(e);
        ev_pending.clear();
        for (auto& q : ev_pairs) { (void)hipEventDestroy(q.a); (void)hipEventDestroy(q.b); }
        ev_pairs.clear();
        for (auto e : ev_free) (void)hipEventDestroy(e);
        ev_free.clear();
        for (auto e : ev_sync_free) (void)hipEventDestroy(e);
        ev_sync_free.clear();
        if (stream) (void)hipStreamDestroy(stream);
        stream = nullptr;
        dev_ready = false;
        last = LastCall();
    }
};

namespace {

template <typename T>
int upload(T** dst, const void* src, size_t bytes, hipStream_t st)
{
    HIP_TRY(hipMalloc((void**)dst, bytes));
    HIP_TRY(hipMemcpyAsync(*dst, src, bytes, hipMemcpyHostToDevice, st));
    return 0;
}


template <int NF, int MODE, int R>
int launch_conv_t(uva_net* n, const ConvArgs& a)
{
    const size_t lds = conv_lds_bytes<NF>(MODE == 0 ? 0 : R);
    auto kfn = conv3x3_kernel<NF, MODE, R>;
    // a net runs at most one trunk (MODE 0) and the u8 / f32 tails (MODE 1 / 2) of its own graph: slots 0-2
    if (!n->attr_set[MODE]) {
        HIP_TRY(hipFuncSetAttribute((const void*)kfn, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds));
        n->attr_set[MODE] = true;
    }
    // 24-feature nets: 160 VGPRs and ~52 KB of LDS per workgroup leave room for two persistent
    // workgroups per CU, whose k-loops and epilogues then overlap (the 64-feature tails fill the
    // register file with weights: one workgroup per CU)
    // (measured at 1080p, 1x HurrDeblur: 1 per CU 1 658 fps, 2 per CU 2 145 fps, 3 per CU 1 905 fps)
    const int per_cu = NF == 24 ? 2 : 1;
    const int grid = std::max(8, (n->ncu / 8) * 8) * per_cu;
    hipLaunchKernelGGL(kfn, dim3(grid), dim3(256), lds, n->stream, a);
    HIP_TRY(hipGetLastError());
    return 0;
}

// trunk layer of the 64-feature nets: split-channel ping-pong kernel, one 8-wave workgroup per CU, 4-row tiles
template <int ABL>
int launch_trunk64_t(uva_net* n, const ConvArgs& a)
{
    const size_t lds = trunk_lds_bytes<64>();
    auto kfn = trunk_kernel<64, ABL>;
    constexpr int SLOT = 3 + (ABL == 0 ? 0 : ABL == 1 ? 1 : ABL == 2 ? 2 : 3);   // slots 3-6
    if (!n->attr_set[SLOT]) {
        HIP_TRY(hipFuncSetAttribute((const void*)kfn, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds));
        n->attr_set[SLOT] = true;
    }
    const int grid = std::max(8, (n->ncu / 8) * 8);
    hipLaunchKernelGGL(kfn, dim3(grid), dim3(512), lds, n->stream, a);
    HIP_TRY(hipGetLastError());
    return 0;
}

int launch_trunk64(uva_net* n, const ConvArgs& a, int ablate = 0)
{
#ifdef UVA_INSTRUMENT
    if (ablate == 1) return launch_trunk64_t<1>(n, a);
    if (ablate == 2) return launch_trunk64_t<2>(n, a);
    if (ablate == 4) return launch_trunk64_t<4>(n, a);
#endif
    (void)ablate;
    return launch_trunk64_t<0>(n, a);
}

template <int NF>
int launch_conv_nf(uva_net* n, int mode, int r, const ConvArgs& a)
{
    if (mode == 0) return launch_conv_t<NF, 0, 1>(n, a);
    if (mode == 1) {
        if (r == 1) return launch_conv_t<NF, 1, 1>(n, a);
        if (r == 2) return launch_conv_t<NF, 1, 2>(n, a);
        if (r == 4) return launch_conv_t<NF, 1, 4>(n, a);
    } else {
        if (r == 1) return launch_conv_t<NF, 2, 1>(n, a);
        if (r == 2) return launch_conv_t<NF, 2, 2>(n, a);
        if (r == 4) return launch_conv_t<NF, 2, 4>(n, a);
    }
    return fail("no kernel for this scale");
}

int launch_conv(uva_net* n, int mode, const ConvArgs& a)
{
    if (n->g.nf == 64) return launch_conv_nf<64>(n, mode, n->g.scale, a);
    if (n->g.nf == 24) return launch_conv_nf<24>(n, mode, n->g.scale, a);
    return fail("no kernel for this trunk width");
}

// u8 tails of the 64-feature 2x and 4x nets: ping-pong kernels on 4-row tiles (the other tails: conv3x3_kernel)
int launch_tail_u8(uva_net* n, const Workspace* ws, ConvArgs ca)
{
    if (n->g.nf == 64 && (n->g.scale == 2 || n->g.scale == 4) && n->layers.back().wpk16) {
        const bool x4 = n->g.scale == 4;
        const size_t lds = x4 ? tail4_lds_bytes<64>() : tail_lds_bytes<64>();
        void (*kfn)(ConvArgs) = x4 ? tail4_kernel<64> : tail_kernel<64, 2>;
        if (!n->attr_set[7]) {
            HIP_TRY(hipFuncSetAttribute((const void*)kfn, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds));
            n->attr_set[7] = true;
        }
        const int grid = std::max(8, (n->ncu / 8) * 8);
        const int per_launch = 8 * (2 * (grid / 8)) * ((TAIL_SCHED_MAX - TRUNK_LOOKAHEAD) / 2 - 1);
        ca.sched4 = ws->d_sched4;
        ca.wpk = n->layers.back().wpk16;
        for (int base = 0; base < ws->ntiles4; base += per_launch) {
            ca.tile_base = base;
            ca.ntiles = std::min(per_launch, ws->ntiles4 - base);
            ca.tiles_per_xcd = (ca.ntiles + 7) / 8;
            hipLaunchKernelGGL(kfn, dim3(grid), dim3(512), lds, n->stream, ca);
            HIP_TRY(hipGetLastError());
        }
        return 0;
    }
    return launch_conv(n, 1, ca);
}

// one trunk layer: the 64-feature nets use the split-channel kernel on 4-row tiles
int launch_trunk(uva_net* n, const Workspace* ws, ConvArgs ca, int ablate = 0)
{
    if (n->g.nf == 64) {
        // a workgroup's tile schedule must fit its LDS table: split very large frames into launches
        const int grid = std::max(8, (n->ncu / 8) * 8);
        const int per_launch = 8 * (2 * (grid / 8)) * ((TRUNK_SCHED_MAX - TRUNK_LOOKAHEAD) / 2 - 1);
        ca.sched4 = ws->d_sched4;
        for (int base = 0; base < ws->ntiles4; base += per_launch) {
            ca.tile_base = base;
            ca.ntiles = std::min(per_launch, ws->ntiles4 - base);
            ca.tiles_per_xcd = (ca.ntiles + 7) / 8;
            if (launch_trunk64(n, ca, ablate)) return 1;
        }
        return 0;
    }
    ca.ntiles = ws->ntiles;
    ca.tiles_per_xcd = (ws->ntiles + 7) / 8;
    return launch_conv(n, 0, ca);
}

// Step lists of trunk2_kernel for one frame geometry: every plane is cut into 30-column strips, a strip
// is a column of 4-row steps walked top to bottom, and the sequence (plane, strip, step) is dealt out to
// the workgroups in contiguous ranges of (nearly) equal length.  A range that ends inside a strip ends a
// SEGMENT there: k producer steps yield 4k - 2 output rows (the consumer needs one intermediate row below
// its last output row), the next segment starts on the following row and recomputes two intermediate rows.
// Consecutive ranges go to the workgroups of one XCD (block b runs on XCD b % 8).
int build_trunk2_schedule(const std::vector<PlaneDesc>& planes, int grid, size_t guard_bytes, std::vector<Trunk2Step>& steps,
                          std::vector<int>& nsteps, int* max_steps, bool narrow_ok = true)
{
    constexpr int PIXB = 128;
    struct Seg { int plane, x0, ya, rows, k; };
    // ranges of equal COST: a step of a narrow strip (<= 14 columns: one fragment column instead of two) runs its k-loops
    // with half the MFMAs and is counted as 8 tenths of a step (measured: its phases are then bounded by the other group's epilogue)
    constexpr int COST = 10, COST_NARROW = 8;
    auto step_cost = [&](const PlaneDesc& p, int x0) { return (narrow_ok && p.w - x0 <= 14) ? COST_NARROW : COST; };
    long long total = 0;
    for (const auto& p : planes)
        for (int x0 = 0; x0 < p.w; x0 += T2_SW) total += (long long)((p.h + 2 + 3) / 4) * step_cost(p, x0);
    std::vector<std::vector<Seg>> per_wg;
    int L = (int)std::max<long long>(4 * COST, (total + grid - 1) / grid);
    for (;; ++L) {
        per_wg.assign(1, {});
        int cap = L;
        auto next_wg = [&]() { per_wg.emplace_back(); cap = L; };
        for (size_t pi = 0; pi < planes.size(); ++pi) {
            const PlaneDesc& p = planes[pi];
            for (int x0 = 0; x0 < p.w; x0 += T2_SW) {
                const int c = step_cost(p, x0);
                int y = 0;
                while (y < p.h) {
                    const int need = (p.h - y + 2 + 3) / 4, fit = cap / c;
                    if (need <= fit) {
                        per_wg.back().push_back({(int)pi, x0, y, p.h - y, need});
                        cap -= need * c;
                        y = p.h;
                    } else if (fit < 2) {
                        next_wg();
                        continue;
                    } else {
                        per_wg.back().push_back({(int)pi, x0, y, 4 * fit - 2, fit});
                        y += 4 * fit - 2;
                        cap = 0;
                    }
                    if (cap < COST_NARROW) next_wg();
                }
            }
        }
        while (!per_wg.empty() && per_wg.back().empty()) per_wg.pop_back();
        if ((int)per_wg.size() <= grid) break;
    }
    int most = 0;
    for (const auto& v : per_wg) {
        int k = 0;
        for (const Seg& sg : v) k += sg.k;
        most = std::max(most, k);
    }
    *max_steps = most;
    const int stride = most + T2_PAD_STEPS;
    steps.assign((size_t)grid * stride, Trunk2Step{make_uint4(0, 0, 0, 0), make_uint4(0, 0, 0, 0)});
    nsteps.assign(grid, 0);
    const int per_xcd = grid / 8;
    for (size_t c = 0; c < per_wg.size(); ++c) {
        const int b = (int)(c % per_xcd) * 8 + (int)(c / per_xcd);
        Trunk2Step* out = steps.data() + (size_t)b * stride;
        int g = 0;
        for (const Seg& sg : per_wg[c]) {
            const PlaneDesc& p = planes[sg.plane];
            const int nb = (sg.rows + 3) / 4;
            for (int j = 0; j < sg.k; ++j, ++g) {
                const int yA = sg.ya - 1 + 4 * j;                        // first intermediate row of the block
                // halo origin = input pixel (yA - 1, x0 - 2) = array position (yA, x0 - 1)
                const long long ao = (long long)guard_bytes +
                                     ((long long)p.act_off + (long long)yA * p.pitch + (sg.x0 - 1)) * PIXB;
                if (ao < 0 || (ao >> 40)) return fail("activation buffer too large for the step encoding");
                unsigned rmask = 0;
                for (int r = 0; r < 4; ++r)
                    if (yA + r >= 0 && yA + r < p.h) rmask |= 1u << r;
                const unsigned c_lo = sg.x0 == 0 ? 1 : 0, c_hi = (unsigned)std::min(32, p.w - sg.x0 + 1);
                // strips of at most 14 columns need only the first of the two 16-column fragment columns (bit 25, both halves)
                const unsigned narrow = (p.w - sg.x0 <= 14 && narrow_ok) ? 1u << 25 : 0u;
                out[g].a = make_uint4((unsigned)ao, (unsigned)(ao >> 32) | (rmask << 8) | (c_lo << 12) | (c_hi << 18) | (1u << 24) | narrow,
                                      (unsigned)(p.pitch * PIXB), (unsigned)sg.plane);
                if (j < nb) {
                    const int yo = sg.ya + 4 * j;
                    const long long bo = (long long)guard_bytes +
                                         ((long long)p.act_off + (long long)(yo + 1) * p.pitch + (sg.x0 + 1)) * PIXB;
                    const unsigned vy = (unsigned)std::min(4, sg.ya + sg.rows - yo), vx = (unsigned)std::min(T2_SW, p.w - sg.x0);
                    out[g].b = make_uint4((unsigned)bo, (unsigned)(bo >> 32) | (vy << 8) | (vx << 11) | (1u << 24) | narrow,
                                          (unsigned)(p.pitch * PIXB), (unsigned)sg.plane);
                }
            }
        }
        nsteps[b] = g;
        for (int k = 0; k < T2_PAD_STEPS; ++k) {      // harmless re-fetches of the last tile, nothing active
            out[g + k].a = out[g - 1].a;
            out[g + k].a.y &= 0xffu;
        }
    }
    return 0;
}

int launch_trunk2(uva_net* n, const Workspace* ws, const Trunk2Args& a)
{
    const size_t lds = trunk2_lds_bytes<64>();
    auto kfn = trunk2_kernel<64>;
    if (!n->attr_set[8]) {
        HIP_TRY(hipFuncSetAttribute((const void*)kfn, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds));
        n->attr_set[8] = true;
    }
    hipLaunchKernelGGL(kfn, dim3(ws->grid2), dim3(512), lds, n->stream, a);
    HIP_TRY(hipGetLastError());
    return 0;
}

// Step lists of trunkw_kernel: the same 30-column strips and 4-row steps as trunk2_kernel's, but a segment that is not its
// workgroup's first starts WITHOUT the two input rows a step shares with the one above it: its first producer step yields
// two valid intermediate rows and its first consumer step nothing, so k steps yield 4 (k - 1) output rows and a segment
// of `rows` output rows beginning at row r0 has its first intermediate block at row r0 - 3 (a workgroup's first segment:
// r0 - 1, 4k - 2 rows -- the kernel's prologue fetches all six rows).  Step g of a segment:
//   producer: intermediate rows yA .. yA+3, yA = r0 - 3 (- 1) + 4g, from the new input rows yA+1 .. yA+4 (+ the two above);
//   consumer: output rows yA-1 .. yA+2, of which those inside the segment are stored, from the last two rows of block g-1
//             and block g.
int build_trunkw_schedule(const std::vector<PlaneDesc>& planes, int grid, size_t guard_bytes, std::vector<Trunk2Step>& steps,
                          std::vector<int>& nsteps, int* max_steps)
{
    constexpr int PIXB = 128;
    // A workgroup's FIRST segment can start with the two input rows a step shares with the one above it (the kernel's prologue
    // fetches and transforms them: entry bit 25) and then yields 4k - 2 rows in k steps instead of 4 (k - 1).  The prologue is
    // 0.3 % of a launch (measured), so this is done only where it shortens the LONGEST list (UVA_TW_SIX: 1 always, 0 never):
    // a whole 1080p frame 69 -> 68 steps (+0.5 %), the reference tiling 73 -> 73 (left alone).  Also measured and not kept
    // (profiles/r04_ab_results.txt block 20): segments at the TOP of a plane starting on two zero rows written by the consumers
    // -- with the six-row starts 73 -> 72 steps at the reference tiling, and the same launch time: a segment's fill step, in
    // which the consumers idle, costs about half a step.
    const char* const sx = uva::debug_env("UVA_TW_SIX");
    const int six_mode = sx ? (std::atoi(sx) != 0 ? 1 : 0) : -1;
    const bool top_ok = false;
    // FOLDED last strips (round 5).  A strip costs its 16 MFMA pair columns whatever its width, and 970 columns (the reference
    // tiling's planes at 1080p) are 32 strips and a THIRD of one: 2 % of a launch's steps compute nothing.  Where two planes have
    // the same size and a last strip of at most TW_FOLD_MAXW = 12 columns, ONE strip walk does both: lanes with pair index 0..7
    // work on the first plane, 8..15 on the second.  Why 12 and not 14: v output columns need the intermediate columns -1..v,
    // i.e. the producers' pairs 0..(v + 1) / 2, and producer pair p reads raw columns 2p..2p+3 -- for v = 13, 14 that is pair 7
    // and raw columns 16, 17, which in a folded step hold the SECOND plane's first pixels (the first version allowed 14: one
    // wrong column per 74-wide plane, 1.4 dB on the 128x96 probe, inside every parity bar -- found by the probe's PSNR moving,
    // now pinned by a byte-for-byte fold on / off test).  Everything in between (rings,
    // transforms, k-loops) is pair-wise and does not care; what differs is where the raw rows come from and where the results
    // go: the second plane's addresses = the first's + a constant (entry .w = that constant - 2048, flags a.y bit 27 / b.y
    // bit 25; csrc/uva_wino.hip.h).  UVA_TW_FOLD=0: off.
    // UVA_TW_FOLD=14 (debug opt-in only) is that first version, kept as the known-bad schedule the tests' structured-error
    // detector must catch (tests/test_gpu_parity.py::test_structure_detector_catches_the_fold14_schedule).
    const char* const sf = uva::debug_env("UVA_TW_FOLD");
    const bool fold_ok = !sf || std::atoi(sf) != 0;
    const int fold_maxw = sf && std::atoi(sf) == 14 ? 14 : TW_FOLD_MAXW;
    std::vector<int> fold_partner(planes.size(), -1);
    std::vector<char> folded_away(planes.size(), 0);
    auto last_x0 = [](const PlaneDesc& p) { return ((p.w + TW_SW - 1) / TW_SW - 1) * TW_SW; };
    if (fold_ok)
        for (size_t i = 0; i < planes.size(); ++i) {
            if (fold_partner[i] >= 0 || folded_away[i] || planes[i].w - last_x0(planes[i]) > fold_maxw) continue;
            for (size_t j = i + 1; j < planes.size(); ++j) {
                if (fold_partner[j] >= 0 || folded_away[j]) continue;
                const long long delta = ((long long)planes[j].act_off - (long long)planes[i].act_off) * PIXB;
                if (planes[j].h == planes[i].h && planes[j].w == planes[i].w && planes[j].pitch == planes[i].pitch && delta > 2048 && delta < (1ll << 31)) {
                    fold_partner[i] = (int)j;
                    folded_away[j] = 1;
                    break;
                }
            }
        }
    struct Seg { int plane, x0, r0, rows, k; bool full, zero; int fold; };
    long long total = 0;
    for (const auto& p : planes) total += (long long)((p.w + TW_SW - 1) / TW_SW) * ((p.h + 3) / 4 + 1);     // (an upper bound: full segments need less)
    auto pack = [&](bool six_ok, std::vector<std::vector<Seg>>& per_wg) {
        int L = (int)std::max<long long>(4, (total + grid - 1) / grid - 2);
        for (;; ++L) {
            per_wg.assign(1, {});
            int cap = L;
            auto next_wg = [&]() { per_wg.emplace_back(); cap = L; };
            for (size_t pi = 0; pi < planes.size(); ++pi) {
                const PlaneDesc& p = planes[pi];
                for (int x0 = 0; x0 < p.w; x0 += TW_SW) {
                    const bool last = x0 + TW_SW >= p.w;
                    if (last && folded_away[pi]) continue;          // done by its partner's last strip
                    const int fold = last ? fold_partner[pi] : -1;
                    int y = 0;
                    while (y < p.h) {
                        const bool zero = top_ok && y == 0;
                        const bool full = zero || (six_ok && per_wg.back().empty() && fold < 0);     // (the prologue's six-row fetch knows one plane)
                        const int need = full ? (p.h - y + 2 + 3) / 4 : (p.h - y + 3) / 4 + 1;
                        if (need <= cap) {
                            per_wg.back().push_back({(int)pi, x0, y, p.h - y, need, full, zero, fold});
                            cap -= need;
                            y = p.h;
                        } else if (cap < 3) {             // a segment of fewer than 3 steps is mostly pipeline fill
                            next_wg();
                            continue;
                        } else {
                            const int rows = full ? 4 * cap - 2 : 4 * (cap - 1);
                            per_wg.back().push_back({(int)pi, x0, y, rows, cap, full, zero, fold});
                            y += rows;
                            cap = 0;
                        }
                        if (cap < 1) next_wg();
                    }
                }
            }
            while (!per_wg.empty() && per_wg.back().empty()) per_wg.pop_back();
            if ((int)per_wg.size() <= grid) break;
        }
        int most = 0;
        for (const auto& v : per_wg) {
            int k = 0;
            for (const Seg& sg : v) k += sg.k;
            most = std::max(most, k);
        }
        return most;
    };
    std::vector<std::vector<Seg>> per_wg;
    if (six_mode >= 0) pack(six_mode != 0, per_wg);
    else {
        std::vector<std::vector<Seg>> with_six;
        const int m0 = pack(false, per_wg), m1 = pack(true, with_six);
        if (m1 < m0) per_wg.swap(with_six);
    }
    int most = 0;
    for (const auto& v : per_wg) {
        int k = 0;
        for (const Seg& sg : v) k += sg.k;
        most = std::max(most, k);
    }
    *max_steps = most;
    const int stride = most + TW_PAD_STEPS;
    steps.assign((size_t)grid * stride, Trunk2Step{make_uint4(0, 0, 0, 0), make_uint4(0, 0, 0, 0)});
    nsteps.assign(grid, 0);
    const int per_xcd = grid / 8;
    for (size_t c = 0; c < per_wg.size(); ++c) {
        const int b = (int)(c % per_xcd) * 8 + (int)(c / per_xcd);
        Trunk2Step* out = steps.data() + (size_t)b * stride;
        int g = 0;
        for (const Seg& sg : per_wg[c]) {
            const PlaneDesc& p = planes[sg.plane];
            for (int j = 0; j < sg.k; ++j, ++g) {
                const int yA = sg.r0 - (sg.full ? 1 : 3) + 4 * j;        // first intermediate row of the block
                // first new input row = pixel (yA + 1, x0 - 2) = array position (yA + 2, x0 - 1)
                const long long ao = (long long)guard_bytes +
                                     ((long long)p.act_off + (long long)(yA + 2) * p.pitch + (sg.x0 - 1)) * PIXB;
                if (ao < 0 || (ao >> 40)) return fail("activation buffer too large for the step encoding");
                unsigned rmask = 0;
                for (int r = 0; r < 4; ++r)
                    if (yA + r >= 0 && yA + r < p.h) rmask |= 1u << r;
                const unsigned c_lo = sg.x0 == 0 ? 1 : 0, c_hi = (unsigned)std::min(32, p.w - sg.x0 + 1);
                // (a folded step carries its first plane's index in .z's top byte: the debug view's, the kernel masks it off)
                if (sg.fold >= 0 && sg.plane > 255) return fail("trunkw schedule: more than 256 planes with folded strips");
                // a folded step: the second plane's pixels lie fold_add + 2048 bytes behind the first's
                const unsigned fold_add = sg.fold >= 0 ? (unsigned)(((long long)planes[sg.fold].act_off - (long long)p.act_off) * PIXB - 2048) : 0u;
                out[g].a = make_uint4((unsigned)ao, (unsigned)(ao >> 32) | (rmask << 8) | (c_lo << 12) | (c_hi << 18) | (1u << 24) |
                                                    ((sg.full && !sg.zero && j == 0) ? 1u << 25 : 0u) | ((sg.zero && j == 0) ? 1u << 26 : 0u) |
                                                    (sg.fold >= 0 ? 1u << 27 : 0u),
                                      (unsigned)(p.pitch * PIXB) | (sg.fold >= 0 ? (unsigned)sg.plane << 24 : 0u),   // (folded: .w is taken, the
                                      sg.fold >= 0 ? fold_add : (unsigned)sg.plane);                                // plane index rides in .z's top byte)
                if ((unsigned)(p.pitch * PIXB) >> 24) return fail("plane too wide for the step encoding");
                // the consumer step stores rows yo + [v0, v1) of its four (yo = yA - 1): those inside the segment
                const int yo = yA - 1;
                const int v0 = std::max(0, sg.r0 - yo), v1 = std::min(4, sg.r0 + sg.rows - yo);
                if (v1 > v0) {
                    const long long bo = (long long)guard_bytes +
                                         ((long long)p.act_off + (long long)(yo + 1) * p.pitch + (sg.x0 + 1)) * PIXB;
                    if (bo < 0 || (bo >> 40)) return fail("activation buffer too large for the step encoding");
                    const unsigned vx = (unsigned)std::min(TW_SW, p.w - sg.x0);
                    out[g].b = make_uint4((unsigned)bo, (unsigned)(bo >> 32) | ((unsigned)v1 << 8) | (vx << 11) | ((unsigned)v0 << 17) | (1u << 24) |
                                                        (sg.fold >= 0 ? 1u << 25 : 0u),
                                          (unsigned)(p.pitch * PIXB) | (sg.fold >= 0 ? (unsigned)sg.plane << 24 : 0u),
                                          sg.fold >= 0 ? fold_add : (unsigned)sg.plane);
                }
            }
        }
        nsteps[b] = g;
        for (int k = 0; k < TW_PAD_STEPS && g > 0; ++k) {     // harmless re-fetches of the last rows, nothing active
            out[g + k].a = out[g - 1].a;
            out[g + k].a.y &= 0xffu;
        }
    }
    return 0;
}

// layers i, i + 1 of the net in one trunkw_kernel launch.  in_negated: the input's channels with a slope above 1 (layer i - 1's)
// are still negated; carry_out: leave layer i + 1's that way for the next launch (its first layer's weights take the sign back)
int launch_trunkw(uva_net* n, const Workspace* ws, TrunkwArgs& a, int i, bool in_negated = false, bool carry_out = false)
{
    for (int k = 0; k < 2; ++k) {
        a.wpk[k] = n->layers[i + k].wpk_w;
        a.bias[k] = n->act16 ? n->layers[i + k].bias_w : n->layers[i + k].bias;
        a.slope[k] = n->layers[i + k].slope;
    }
    if (in_negated) a.wpk[0] = n->layers[i].wpk_wn;
    const int act = !n->act16 ? TW_ACT_F32 : (n->layers[i + 1].flip_w && !carry_out) ? TW_ACT_F16_FLIP : TW_ACT_F16;
    HIP_TRY(launch_trunkw_kernel(n->stream, ws->grid2, a, act));
    return 0;
}

// Row descriptors of sub10_kernel for an h x w frame: 60-column strips, the sequence (strip, row) dealt out to the
// workgroups in contiguous ranges; every range (segment) starts 10 rows early and ends 9 rows late (the rows the
// layers in between need), only its own rows are written out.
// `frames` frames of the one geometry in one launch: the sequence runs over (frame, strip, row); a row's frame sits in z >> 8.
int build_sub10_rows(int h, int w, int frames, int grid, std::vector<uint4>& rows, std::vector<int>& nrows, int* max_rows)
{
    if (frames < 1 || frames > S10_MAXB || h > S10_MAX_H) return 2;
    const int ns1 = (w + S10_VALID - 1) / S10_VALID;
    const int ns = ns1 * frames;             // k = frame * ns1 + strip
    const long long total = (long long)ns * h;
    struct Seg { int k, y0, n; };
    std::vector<std::vector<Seg>> per_wg;
    int D = (int)std::max<long long>(2 * S10_NL + 4, (total + grid - 1) / grid + 2 * S10_NL);
    for (;; ++D) {
        per_wg.assign(1, {});
        int cap = D;
        for (int k = 0; k < ns; ++k) {
            int y = 0;
            while (y < h) {
                if (cap < 2 * S10_NL + 1) { per_wg.emplace_back(); cap = D; }
                const int n = std::min(h - y, cap - 2 * S10_NL);
                per_wg.back().push_back({k, y, n});
                cap -= n + 2 * S10_NL;
                y += n;
            }
        }
        if ((int)per_wg.size() <= grid) break;
    }
    if (D > S10_MAX_ROWS) return 2;      // row table does not fit the kernel's LDS copy: the caller takes the per-pair path
    *max_rows = D;
    rows.assign((size_t)grid * D, make_uint4(0, 0, 0, 0));
    nrows.assign(grid, 0);
    const int per_xcd = grid / 8;
    for (size_t c = 0; c < per_wg.size(); ++c) {
        const int b = (int)(c % per_xcd) * 8 + (int)(c / per_xcd);
        uint4* out = rows.data() + (size_t)b * D;
        int g = 0;
        for (const Seg& sg : per_wg[c])
            for (int y = sg.y0 - S10_NL; y < sg.y0 + sg.n + S10_NL; ++y, ++g) {
                // w = how many rows y lies outside the segment's own rows (0 inside, 1..10): layer s (0 = the first) is needed on rows
                // with w <= 9 - s only, and the kernel's wave of that layer skips the others
                const int dist = y < sg.y0 ? sg.y0 - y : y >= sg.y0 + sg.n ? y - (sg.y0 + sg.n - 1) : 0;
                out[g] = make_uint4((unsigned)y, (unsigned)((sg.k % ns1) * S10_VALID - S10_NL), (dist == 0 ? 1u : 0u) | ((unsigned)(sg.k / ns1) << 8),
                                    (unsigned)dist);
            }
        nrows[b] = g;
    }
    return 0;
}

// Row descriptors of sub5_kernel (both launches use the same lists): PAIRS of 54-column strips, the sequence (pair, row) dealt
// out to the workgroups in contiguous ranges; every range (segment) starts 5 rows early and ends 5 rows late -- what five
// 3x3 layers need --, only its own rows are written out.  x = plane row, y = plane column of computed column 0 of the pair's
// FIRST strip (the second one's is S5_VALID further right), z = 1: written out.
int build_sub5_rows(int h, int w, int grid, std::vector<uint4>& rows, std::vector<int>& nrows, int* max_rows)
{
    const int np = (w + S5_PAIRW - 1) / S5_PAIRW;
    const long long total = (long long)np * h;
    struct Seg { int k, y0, n; };
    std::vector<std::vector<Seg>> per_wg;
    int D = (int)std::max<long long>(2 * S5_NL + 4, (total + grid - 1) / grid + 2 * S5_NL);
    for (;; ++D) {
        per_wg.assign(1, {});
        int cap = D;
        for (int k = 0; k < np; ++k) {
            int y = 0;
            while (y < h) {
                if (cap < 2 * S5_NL + 1) { per_wg.emplace_back(); cap = D; }
                const int n = std::min(h - y, cap - 2 * S5_NL);
                per_wg.back().push_back({k, y, n});
                cap -= n + 2 * S5_NL;
                y += n;
            }
        }
        if ((int)per_wg.size() <= grid) break;
    }
    if (D > S5_MAX_ROWS) return 2;       // the row table does not fit the kernel's LDS copy: the caller takes another path
    *max_rows = D;
    rows.assign((size_t)grid * D, make_uint4(0, 0, 0, 0));
    nrows.assign(grid, 0);
    const int per_xcd = grid / 8;
    for (size_t c = 0; c < per_wg.size(); ++c) {
        const int b = (int)(c % per_xcd) * 8 + (int)(c / per_xcd);
        uint4* out = rows.data() + (size_t)b * D;
        int g = 0;
        for (const Seg& sg : per_wg[c])
            for (int y = sg.y0 - S5_NL; y < sg.y0 + sg.n + S5_NL; ++y, ++g)
                out[g] = make_uint4((unsigned)y, (unsigned)(sg.k * S5_PAIRW - S5_NL), (y >= sg.y0 && y < sg.y0 + sg.n) ? 1u : 0u, 0u);
        nrows[b] = g;
    }
    return 0;
}

// the 24-feature 1x net as two launches of five layers (u8 route, one plane); returns 2 when the frame is too large for it
int launch_sub5(uva_net* n, Workspace* ws, const void* src, size_t src_stride, void* dst, size_t dst_stride,
                unsigned long long* dbg = nullptr, int dbg_part = -1)
{
    if (ws->sub5_unfit) return 2;
    if (!ws->d_rows5) {
        std::vector<uint4> rows;
        std::vector<int> nrows;
        ws->grid5 = std::max(8, (n->ncu / 8) * 8);
        if (build_sub5_rows(ws->h, ws->w, ws->grid5, rows, nrows, &ws->max_rows5)) {
            ws->sub5_unfit = true;
            return 2;
        }
        // into locals first: a failure half way leaves nothing in the workspace (the next call starts over instead of leaking)
        char* d_mid = nullptr;
        uint4* d_rows = nullptr;
        int* d_nrows = nullptr;
        hipError_t e = hipMalloc((void**)&d_mid, (size_t)ws->h * ws->w * S5_MIDB);
        if (e == hipSuccess) e = hipMalloc((void**)&d_rows, rows.size() * sizeof(uint4));
        if (e == hipSuccess) e = hipMalloc((void**)&d_nrows, nrows.size() * sizeof(int));
        if (e == hipSuccess) e = hipMemcpyAsync(d_rows, rows.data(), rows.size() * sizeof(uint4), hipMemcpyHostToDevice, n->stream);
        if (e == hipSuccess) e = hipMemcpyAsync(d_nrows, nrows.data(), nrows.size() * sizeof(int), hipMemcpyHostToDevice, n->stream);
        if (e == hipSuccess) e = hipStreamSynchronize(n->stream);
        if (e != hipSuccess) {
            if (d_mid) (void)hipFree(d_mid);
            if (d_rows) (void)hipFree(d_rows);
            if (d_nrows) (void)hipFree(d_nrows);
            return fail(std::string("sub5 workspace: ") + hipGetErrorString(e));
        }
        ws->d_mid5 = d_mid; ws->d_rows5 = d_rows; ws->d_nrows5 = d_nrows;
    }
    for (int part = 0; part < 2; ++part) {
        Sub5Args a;
        std::memset(&a, 0, sizeof a);
        a.src = (const uint8_t*)src; a.src_stride = src_stride;
        a.dst = (uint8_t*)dst; a.dst_stride = dst_stride;
        a.mid = ws->d_mid5;
        a.h = ws->h; a.w = ws->w;
        a.rows = ws->d_rows5; a.nrows = ws->d_nrows5; a.max_rows = ws->max_rows5;
        a.dbg = part == dbg_part ? dbg : nullptr;
        for (int i = 0; i < S5_NL; ++i) {
            a.wpk[i] = n->layers[S5_NL * part + i].wpk_s10;
            a.bias[i] = n->layers[S5_NL * part + i].bias_s10;
            a.slope[i] = n->layers[S5_NL * part + i].slope;
        }
        HIP_TRY(launch_sub5_kernel(n->stream, ws->grid5, a, part));
    }
    return 0;
}

// the whole 24-feature 1x net in one launch (u8 route, one plane per frame), `frames` frames of the workspace's geometry at once;
// returns 2 when that many frames do not fit the kernel's row table (the caller takes fewer, or for one frame another path)
int launch_sub10(uva_net* n, Workspace* ws, const void* const* srcs, size_t src_stride, void* const* dsts, size_t dst_stride, int frames,
                 unsigned long long* dbg = nullptr)
{
    if (ws->sub10_unfit || frames < 1 || frames > ws->sub10_max_batch) return 2;
    const int bi = frames - 1;
    if (!ws->d_rows10[bi]) {
        std::vector<uint4> rows;
        std::vector<int> nrows;
        ws->grid10 = std::max(8, (n->ncu / 8) * 8);
        if (build_sub10_rows(ws->h, ws->w, frames, ws->grid10, rows, nrows, &ws->max_rows10[bi])) {
            if (frames == 1) ws->sub10_unfit = true;
            ws->sub10_max_batch = frames - 1;
            return 2;
        }
        uint4* d_rows = nullptr;
        int* d_nrows = nullptr;
        hipError_t e = hipMalloc((void**)&d_rows, rows.size() * sizeof(uint4));
        if (e == hipSuccess) e = hipMalloc((void**)&d_nrows, nrows.size() * sizeof(int));
        if (e == hipSuccess) e = hipMemcpyAsync(d_rows, rows.data(), rows.size() * sizeof(uint4), hipMemcpyHostToDevice, n->stream);
        if (e == hipSuccess) e = hipMemcpyAsync(d_nrows, nrows.data(), nrows.size() * sizeof(int), hipMemcpyHostToDevice, n->stream);
        if (e == hipSuccess) e = hipStreamSynchronize(n->stream);
        if (e != hipSuccess) {                 // nothing half-made stays in the workspace
            if (d_rows) (void)hipFree(d_rows);
            if (d_nrows) (void)hipFree(d_nrows);
            return fail(std::string("sub10 row table: ") + hipGetErrorString(e));
        }
        ws->d_rows10[bi] = d_rows;
        ws->d_nrows10[bi] = d_nrows;
    }
    Sub10Args a;
    std::memset(&a, 0, sizeof a);
    for (int f = 0; f < S10_MAXB; ++f) {       // (unused slots repeat the last frame: never addressed, never null)
        a.src[f] = (const uint8_t*)srcs[std::min(f, frames - 1)];
        a.dst[f] = (uint8_t*)dsts[std::min(f, frames - 1)];
    }
    a.src_stride = src_stride;
    a.dst_stride = dst_stride;
    a.h = ws->h; a.w = ws->w;
    a.rows = ws->d_rows10[bi]; a.nrows = ws->d_nrows10[bi]; a.max_rows = ws->max_rows10[bi];
    a.dbg = dbg;
    for (int i = 0; i < S10_NL; ++i) {
        a.wpk[i] = n->layers[i].wpk_s10;
        a.bias[i] = n->layers[i].bias_s10;
        a.slope[i] = n->layers[i].slope;
    }
    HIP_TRY(launch_sub10_kernel(n->stream, ws->grid10, a));
    return 0;
}
int launch_sub10(uva_net* n, Workspace* ws, const void* src, size_t src_stride, void* dst, size_t dst_stride, unsigned long long* dbg = nullptr)
{
    return launch_sub10(n, ws, &src, src_stride, &dst, dst_stride, 1, dbg);
}

// two trunk layers of the 24-feature net per launch (pair24_kernel), two persistent workgroups per CU
int launch_pair24(uva_net* n, const ConvArgs& a)
{
    const size_t lds = pair24_lds_bytes();
    if (!n->attr_set[9]) {
        HIP_TRY(hipFuncSetAttribute((const void*)pair24_kernel, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds));
        n->attr_set[9] = true;
    }
    const int grid = std::max(8, (n->ncu / 8) * 8) * 2;
    hipLaunchKernelGGL(pair24_kernel, dim3(grid), dim3(256), lds, n->stream, a);
    HIP_TRY(hipGetLastError());
    return 0;
}

int launch_head(uva_net* n, bool f32, const HeadArgs& a)
{
    // headp_kernel (persistent, the next tile's pixels requested while this one is computed) unless UVA_HEAD_PERSIST=0
    const char* const hp = uva::debug_env("UVA_HEAD_PERSIST");        // (read per call: the GPU test switches it between two frames)
    const bool persist = !hp || std::atoi(hp) != 0;
    if (persist && a.sink) {
        const dim3 grid(std::min(a.ntiles, n->ncu * HEADP_WG_PER_CU)), block(256);
        if (n->g.nf == 64) {
            const size_t lds = headp_lds_bytes<64>();
            if (f32) hipLaunchKernelGGL((headp_kernel<64, 1>), grid, block, lds, n->stream, a);
            else hipLaunchKernelGGL((headp_kernel<64, 0>), grid, block, lds, n->stream, a);
        } else {
            const size_t lds = headp_lds_bytes<24>();
            if (f32) hipLaunchKernelGGL((headp_kernel<24, 1>), grid, block, lds, n->stream, a);
            else hipLaunchKernelGGL((headp_kernel<24, 0>), grid, block, lds, n->stream, a);
        }
        HIP_TRY(hipGetLastError());
        return 0;
    }
    const dim3 grid(a.ntiles), block(256);
    if (n->g.nf == 64) {
        if (f32) hipLaunchKernelGGL((head_kernel<64, 1>), grid, block, 0, n->stream, a);
        else hipLaunchKernelGGL((head_kernel<64, 0>), grid, block, 0, n->stream, a);
    } else {
        if (f32) hipLaunchKernelGGL((head_kernel<24, 1>), grid, block, 0, n->stream, a);
        else hipLaunchKernelGGL((head_kernel<24, 0>), grid, block, 0, n->stream, a);
    }
    HIP_TRY(hipGetLastError());
    return 0;
}

int ensure_device(uva_net* n)
{
    if (n->generic ? !(n->gg.param_loaded && n->gg.model_loaded) : !(n->g.param_loaded && n->g.model_loaded))
        return fail("net has no model: call load_param and load_model first");
    if (n->dev_ready) {
        HIP_TRY(hipSetDevice(n->device));
        return 0;
    }
    int count = 0;
    if (hipGetDeviceCount(&count) != hipSuccess || count <= 0)
        return fail("no HIP device available: libuva has no CPU path");
    if (n->device < 0 || n->device >= count)
        return fail("HIP device " + std::to_string(n->device) + " does not exist (" + std::to_string(count) + " visible)");
    HIP_TRY(hipSetDevice(n->device));
    hipDeviceProp_t prop;
    HIP_TRY(hipGetDeviceProperties(&prop, n->device));
    if (std::strncmp(prop.gcnArchName, "gfx950", 6) != 0)
        return fail(std::string("libuva is built for gfx950 (MI355X) only; device is ") + prop.gcnArchName);
    n->ncu = prop.multiProcessorCount;
    // dev_ready is set only once every upload below has succeeded; any failure on the way releases what
    // was allocated (free_device works on partial state), so a later call starts from scratch
    n->dev_partial = true;
    struct Undo {
        uva_net* n;
        ~Undo() { if (!n->dev_ready) n->free_device(); }
    } undo{n};
    HIP_TRY(hipStreamCreateWithFlags(&n->stream, hipStreamNonBlocking));
    if (const char* e = uva::debug_env("UVA_TRUNK_FUSION")) n->fuse_pairs = std::atoi(e) != 0;
    if (const char* e = uva::debug_env("UVA_TRUNK_WINO")) n->wino = std::atoi(e) != 0;
    if (const char* e = uva::debug_env("UVA_TW_ACT16")) n->act16 = std::atoi(e) != 0;
    if (const char* e = uva::debug_env("UVA_TW_CARRY")) n->carry = std::atoi(e) != 0;
    if (const char* e = uva::debug_env("UVA_SUB10")) n->fuse_all = std::atoi(e) != 0;
    if (const char* e = uva::debug_env("UVA_SUB5")) n->split5 = std::atoi(e) != 0;
    if (const char* e = uva::debug_env("UVA_GENERIC_LDS")) n->generic_lds_conv = std::atoi(e) != 0;
    if (const char* e = uva::debug_env("UVA_GENERIC_FUSE_ADD")) n->generic_fuse_add = std::atoi(e) != 0;
    HIP_TRY(hipMalloc((void**)&n->d_sink, 64 * 128 + 256));
    if (n->generic) {
        // generic graph: every convolution's MFMA image ([tap][cin/32][cout/16][lane][8]) and padded bias
        const GenericGraph& gg = n->gg;
        n->gd.convs.resize(gg.convs.size());
        for (const GLayer& gl : gg.layers) {
            if (gl.kind != GLayer::CONV) continue;
            const ConvWeights& c = gg.convs[gl.conv];
            GenericDevice::ConvDev& cd = n->gd.convs[gl.conv];
            cd.cin_pad = GenericDevice::pad32(c.cin);
            cd.cout_pad = (c.cout + 15) / 16 * 16;
            std::vector<uint16_t> pk;
            pack_generic(c, gl.ksize, cd.cin_pad, cd.cout_pad, pk);
            if (upload(&cd.wpk, pk.data(), pk.size() * 2, n->stream)) return 1;
            if (cd.cout_pad <= 64 && cd.cin_pad <= 192 && g_conv3_lds_bytes(cd.cin_pad, cd.cout_pad / 16, gl.ksize) <= 160 * 1024) {
                std::vector<uint16_t> pkl;
                pack_generic(c, gl.ksize, cd.cin_pad, cd.cout_pad, pkl, true);
                if (upload(&cd.wpk_lds, pkl.data(), pkl.size() * 2, n->stream)) return 1;
                HIP_TRY(hipStreamSynchronize(n->stream));
            }
            if (gl.ksize == 3 && cd.cin_pad == 192 && cd.cout_pad == 64) {           // a dense block's last convolution: g_conv3_sww
                std::vector<uint16_t> pkw;
                pack_generic_wino(c, cd.cin_pad, cd.cout_pad, pkw);
                if (upload(&cd.wpk_w, pkw.data(), pkw.size() * 2, n->stream)) return 1;
                HIP_TRY(hipStreamSynchronize(n->stream));
            }
            std::vector<float> b((size_t)cd.cout_pad, 0.f);
            std::copy(c.bias.begin(), c.bias.end(), b.begin());
            if (upload(&cd.bias, b.data(), b.size() * 4, n->stream)) return 1;
            HIP_TRY(hipStreamSynchronize(n->stream));
        }
        n->gd.rdbs = find_rdbs(gg);
        n->gd.prelu.assign(gg.prelu.size(), nullptr);
        for (size_t i = 0; i < gg.prelu.size(); ++i) {
            if (upload(&n->gd.prelu[i], gg.prelu[i].data(), gg.prelu[i].size() * 4, n->stream)) return 1;
            HIP_TRY(hipStreamSynchronize(n->stream));
        }
        n->dev_ready = true;
        return 0;
    }
    // weights: head, trunk..., tail
    const Graph& g = n->g;
    n->layers.resize(g.convs.size());
    for (size_t i = 0; i < g.convs.size(); ++i) {
        std::vector<uint16_t> pk;
        int mf = 0;
        if (i == 0) pack_head(g.convs[i], pk, &mf);
        else if (g.nf == 64 && i + 1 < g.convs.size()) { pack_trunk64(g.convs[i], pk); mf = 2; }   // trunk_kernel<64>
        else pack_conv3x3(g.convs[i], g.nf, pk, nullptr, &mf);
        DeviceLayer& dl = n->layers[i];
        if (upload(&dl.wpk, pk.data(), pk.size() * 2, n->stream)) return 1;
        std::vector<uint16_t> pkw;
        if (g.nf == 64 && i > 0 && i + 1 < g.convs.size()) {
            // Pairs are (1, 2), (3, 4), ...: an odd layer is the first of its launch.  With the PReLU on packed fp16 a channel
            // whose slope exceeds 1 is computed negated (uva_wino.h): the first layer's stay so on their way through LDS into
            // the second layer (in_sign), the second layer's get their sign back in front of the stores.
            std::vector<float> sin(64, 1.f), sout(64, 1.f), bw(64, 0.f);
            if (n->act16) {
                if (!(i & 1))
                    for (int c = 0; c < 64; ++c) sin[c] = g.slopes[i - 1][c] > 1.f ? -1.f : 1.f;
                for (int c = 0; c < 64; ++c) {
                    sout[c] = g.slopes[i][c] > 1.f ? -1.f : 1.f;
                    dl.flip_w = dl.flip_w || sout[c] < 0.f;
                }
            }
            for (int c = 0; c < 64; ++c) bw[c] = sout[c] * g.convs[i].bias[c];
            pack_trunk64_wino(g.convs[i], pkw, sin.data(), sout.data());
            if (upload(&dl.wpk_w, pkw.data(), pkw.size() * 2, n->stream)) return 1;
            if (upload(&dl.bias_w, bw.data(), bw.size() * 4, n->stream)) return 1;
            HIP_TRY(hipStreamSynchronize(n->stream));
            if (n->act16 && (i & 1) && i >= 3) {
                // first layer of a launch behind another launch: a second image for an input whose flipped channels arrive negated
                bool any = false;
                for (int c = 0; c < 64; ++c) {
                    sin[c] = g.slopes[i - 1][c] > 1.f ? -1.f : 1.f;
                    any = any || sin[c] < 0.f;
                }
                if (any) {
                    pack_trunk64_wino(g.convs[i], pkw, sin.data(), sout.data());
                    if (upload(&dl.wpk_wn, pkw.data(), pkw.size() * 2, n->stream)) return 1;
                    HIP_TRY(hipStreamSynchronize(n->stream));
                }
            }
        }
        std::vector<uint16_t> pk16;
        if (g.nf == 64 && (g.scale == 2 || g.scale == 4) && i + 1 == g.convs.size()) {
            pack_tail64(g.convs[i], pk16);
            if (upload(&dl.wpk16, pk16.data(), pk16.size() * 2, n->stream)) return 1;
        }
        if (g.nf == 24 && g.scale == 1 && g.convs.size() == (size_t)S10_NL) {
            // channels whose PReLU slope exceeds 1 travel negated between the layers (uva_model.h pack_sub16)
            std::vector<float> sin(g.convs[i].cin, 1.f), sout(g.convs[i].cout, 1.f), bs(32, 0.f);
            if (i > 0)
                for (int c = 0; c < g.convs[i].cin; ++c) sin[c] = g.slopes[i - 1][c] > 1.f ? -1.f : 1.f;
            if (i + 1 < g.convs.size())
                for (int c = 0; c < g.convs[i].cout; ++c) sout[c] = g.slopes[i][c] > 1.f ? -1.f : 1.f;
            const int brow = g.convs[i].cout == 3 ? 4 : 1;       // (the last layer's channel j sits in MFMA row 4j: pack_sub16)
            for (int c = 0; c < g.convs[i].cout; ++c) bs[brow * c] = sout[c] * g.convs[i].bias[c];
            std::vector<uint16_t> pks;
            pack_sub16(g.convs[i], pks, nullptr, nullptr, sin.data(), sout.data());
            if (upload(&dl.wpk_s10, pks.data(), pks.size() * 2, n->stream)) return 1;
            if (upload(&dl.bias_s10, bs.data(), bs.size() * 4, n->stream)) return 1;
            HIP_TRY(hipStreamSynchronize(n->stream));
        }
        std::vector<float> b((size_t)mf * 32, 0.f), s((size_t)mf * 32, 0.f);
        std::copy(g.convs[i].bias.begin(), g.convs[i].bias.end(), b.begin());
        if (upload(&dl.bias, b.data(), b.size() * 4, n->stream)) return 1;
        if (i + 1 < g.convs.size()) {
            std::copy(g.slopes[i].begin(), g.slopes[i].end(), s.begin());
            if (upload(&dl.slope, s.data(), s.size() * 4, n->stream)) return 1;
        }
        HIP_TRY(hipStreamSynchronize(n->stream));   // pk / b / s go out of scope
    }
    n->dev_ready = true;
    return 0;
}

// Plane list of one frame: the reference's tile grid (upscale_processing.py:398-434, :499-516)
// or a single whole-frame plane (apply_model, :263-288).
int build_planes(int h, int w, int tile_size, int border, std::vector<PlaneDesc>& out)
{
    out.clear();
    auto add = [&](int sy0, int sx0, int ph, int pw, int cy0, int cy1, int cx0, int cx1) {
        PlaneDesc p;
        std::memset(&p, 0, sizeof p);
        p.h = ph; p.w = pw;
        p.src_y0 = sy0; p.src_x0 = sx0;
        p.core_y0 = cy0; p.core_y1 = cy1; p.core_x0 = cx0; p.core_x1 = cx1;
        out.push_back(p);
    };
    if (tile_size <= 0) {
        add(0, 0, h, w, 0, h, 0, w);
    } else {
        const int tiles_x = (w + tile_size - 1) / tile_size, tiles_y = (h + tile_size - 1) / tile_size;
        if ((long long)tiles_x * tiles_y > MAX_PLANES) return fail("frame needs more than 64 tiles");
        for (int ty = 0; ty < tiles_y; ++ty)
            for (int tx = 0; tx < tiles_x; ++tx) {
                const int y0 = ty * tile_size, x0 = tx * tile_size;
                const int y1 = std::min(y0 + tile_size, h), x1 = std::min(x0 + tile_size, w);
                const int by0 = y0 >= border ? border : 0, by1 = y1 <= h - border ? border : 0;
                const int bx0 = x0 >= border ? border : 0, bx1 = x1 <= w - border ? border : 0;
                add(y0 - by0, x0 - bx0, (y1 + by1) - (y0 - by0), (x1 + bx1) - (x0 - bx0), by0, by0 + (y1 - y0),
                    bx0, bx0 + (x1 - x0));
            }
    }
    return 0;
}

// Work-tile counts, activation pitch and array offset of every plane (see PlaneDesc).
void layout_planes(std::vector<PlaneDesc>& planes, size_t* pixels, int* ntiles, int* ntiles4)
{
    size_t pix = 0;
    int tiles = 0, tiles4 = 0;
    for (auto& p : planes) {
        p.nty = (p.h + TH - 1) / TH;
        p.ntx = (p.w + TW - 1) / TW;
        p.pitch = p.ntx * TW + 2;
        p.tile_begin = tiles;
        p.nty4 = (p.h + 3) / 4;
        p.tile_begin4 = tiles4;
        p.act_off = (long long)pix;
        tiles += p.nty * p.ntx;
        tiles4 += p.nty4 * p.ntx;
        pix += (size_t)(p.nty * TH + 2) * p.pitch;
    }
    if (pixels) *pixels = pix;
    if (ntiles) *ntiles = tiles;
    if (ntiles4) *ntiles4 = tiles4;
}

int get_workspace(uva_net* n, int h, int w, int tile_size, int border, Workspace** out)
{
    if (tile_size <= 0) { tile_size = 0; border = 0; }
    for (auto it = n->wss.begin(); it != n->wss.end(); ++it)
        if (it->h == h && it->w == w && it->tile_size == tile_size && it->border == border) {
            n->wss.splice(n->wss.begin(), n->wss, it);
            *out = &n->wss.front();
            return 0;
        }
    Workspace ws;
    ws.h = h; ws.w = w; ws.tile_size = tile_size; ws.border = border;
    if (build_planes(h, w, tile_size, border, ws.planes)) return 1;
    size_t pix = 0;
    int tiles = 0, tiles4 = 0;
    layout_planes(ws.planes, &pix, &tiles, &tiles4);
    ws.ntiles = tiles;
    ws.ntiles4 = tiles4;
    ws.act_pixels = pix;
    // everything that can be refused is checked BEFORE anything is allocated
    std::vector<uint4> sched4;
    std::vector<Trunk2Step> steps2, stepsw;
    std::vector<int> nsteps2, nstepsw;
    int max_pitch = 0;
    for (auto& p : ws.planes) max_pitch = std::max(max_pitch, p.pitch);
    ws.guard_bytes = (size_t)8 * max_pitch * n->g.nf * 2;
    if (n->g.nf == 64) {
        using G4 = Geo<64, 4>;
        sched4.reserve((size_t)tiles4);
        for (size_t pi = 0; pi < ws.planes.size(); ++pi) {
            const PlaneDesc& p = ws.planes[pi];
            if (p.nty4 >= 4096 || p.ntx >= 256) return fail("frame too large for the tile schedule encoding");
            for (int ty = 0; ty < p.nty4; ++ty)
                for (int tx = 0; tx < p.ntx; ++tx) {
                    const unsigned long long off =
                        ((unsigned long long)p.act_off + (unsigned long long)(ty * 4) * p.pitch + (unsigned long long)tx * TW) * G4::PIXB;
                    if (off >> 40) return fail("activation buffer too large for the tile schedule encoding");
                    const int vy = std::min(4, p.h - ty * 4), vx = std::min(TW, p.w - tx * TW);
                    sched4.push_back(make_uint4((unsigned)off, (unsigned)(off >> 32) | ((unsigned)pi << 8), (unsigned)(p.pitch * G4::PIXB),
                                                (unsigned)(vx | (vy << 6) | (tx << 9) | (ty << 17))));
                }
        }
        ws.grid2 = std::max(8, (n->ncu / 8) * 8);
        const char* const nv = uva::debug_env("UVA_T2_NARROW");      // (A/B switch: 0 = every strip computes both fragment columns)
        if (build_trunk2_schedule(ws.planes, ws.grid2, ws.guard_bytes, steps2, nsteps2, &ws.max_steps2, !(nv && std::atoi(nv) == 0))) return 1;
        if (build_trunkw_schedule(ws.planes, ws.grid2, ws.guard_bytes, stepsw, nstepsw, &ws.max_stepsw)) return 1;
    }
    const size_t bytes = pix * (size_t)n->g.nf * 2 + 2 * ws.guard_bytes;
    // LRU over the cached geometries, by bytes (the fused route keeps one workspace per frame size, the
    // float / Extractor route one per distinct tile shape -- 9 for a 4000x2200 frame): plenty fit 288 GB
    constexpr size_t CACHE_BYTES = (size_t)96 << 30;
    constexpr size_t CACHE_ENTRIES = 32;
    auto cached_bytes = [&]() {
        size_t t = 0;
        for (auto& w2 : n->wss) t += w2.alloc_bytes;
        return t;
    };
    while (!n->wss.empty() && (n->wss.size() >= CACHE_ENTRIES || cached_bytes() + 2 * bytes > CACHE_BYTES)) {
        HIP_TRY(hipStreamSynchronize(n->stream));
        if (n->last.ws == &n->wss.back()) n->last = LastCall();
        n->wss.back().release();
        n->wss.pop_back();
    }
    // from here on a failure releases what this workspace already holds
    struct Guard {
        Workspace* w;
        ~Guard() { if (w) w->release(); }
    } guard{&ws};
    for (int i = 0; i < 2; ++i) {
        HIP_TRY(hipMalloc((void**)&ws.act_base[i], bytes));
        HIP_TRY(hipMemsetAsync(ws.act_base[i], 0, bytes, n->stream));   // the zero border lives here forever
        ws.act[i] = (_Float16*)(ws.act_base[i] + ws.guard_bytes);
    }
    ws.alloc_bytes = 2 * bytes;
    HIP_TRY(hipMalloc((void**)&ws.d_planes, ws.planes.size() * sizeof(PlaneDesc)));
    HIP_TRY(hipMemcpyAsync(ws.d_planes, ws.planes.data(), ws.planes.size() * sizeof(PlaneDesc),
                           hipMemcpyHostToDevice, n->stream));
    if (n->g.nf == 64) {
        HIP_TRY(hipMalloc((void**)&ws.d_sched4, sched4.size() * sizeof(uint4)));
        HIP_TRY(hipMemcpyAsync(ws.d_sched4, sched4.data(), sched4.size() * sizeof(uint4), hipMemcpyHostToDevice, n->stream));
        HIP_TRY(hipMalloc((void**)&ws.d_steps2, steps2.size() * sizeof(Trunk2Step)));
        HIP_TRY(hipMalloc((void**)&ws.d_nsteps2, nsteps2.size() * sizeof(int)));
        HIP_TRY(hipMemcpyAsync(ws.d_steps2, steps2.data(), steps2.size() * sizeof(Trunk2Step), hipMemcpyHostToDevice, n->stream));
        HIP_TRY(hipMemcpyAsync(ws.d_nsteps2, nsteps2.data(), nsteps2.size() * sizeof(int), hipMemcpyHostToDevice, n->stream));
        HIP_TRY(hipMalloc((void**)&ws.d_stepsw, stepsw.size() * sizeof(Trunk2Step)));
        HIP_TRY(hipMalloc((void**)&ws.d_nstepsw, nstepsw.size() * sizeof(int)));
        HIP_TRY(hipMemcpyAsync(ws.d_stepsw, stepsw.data(), stepsw.size() * sizeof(Trunk2Step), hipMemcpyHostToDevice, n->stream));
        HIP_TRY(hipMemcpyAsync(ws.d_nstepsw, nstepsw.data(), nstepsw.size() * sizeof(int), hipMemcpyHostToDevice, n->stream));
    }
    HIP_TRY(hipStreamSynchronize(n->stream));
    guard.w = nullptr;
    n->wss.push_front(ws);
    *out = &n->wss.front();
    return 0;
}

hipEvent_t take_event(uva_net* n);
hipEvent_t take_sync_event(uva_net* n);

// A pair of events around one launch (profiling): begin() records the first, end() the second and queues the pair; a pair
// that is not completed -- an event could not be created, a call in between failed -- gives its events back.
struct EvPairScope {
    uva_net* n;
    uva_net::EvPair p;
    bool queued = false;
    EvPairScope(uva_net* net, int kind, bool on) : n(net), p{nullptr, nullptr, kind}
    {
        if (!on) return;
        p.a = take_event(n);
        p.b = p.a ? take_event(n) : nullptr;
        if (p.a && !p.b) { n->ev_free.push_back(p.a); p.a = nullptr; }
    }
    hipError_t begin() { return p.b ? hipEventRecord(p.a, n->stream) : hipSuccess; }
    hipError_t end()
    {
        if (!p.b) return hipSuccess;
        const hipError_t e = hipEventRecord(p.b, n->stream);
        if (e == hipSuccess) { n->ev_pairs.push_back(p); queued = true; }
        return e;
    }
    ~EvPairScope()
    {
        if (queued) return;
        if (p.a) n->ev_free.push_back(p.a);
        if (p.b) n->ev_free.push_back(p.b);
    }
};

// -> nullptr (with the error recorded) if the runtime cannot create another event
hipEvent_t take_event(uva_net* n)
{
    if (!n->ev_free.empty()) {
        hipEvent_t e = n->ev_free.back();
        n->ev_free.pop_back();
        return e;
    }
    // (hipEventDisableSystemFence -- no L2 write-back at a timing event -- measured: no difference in the frame time or in the
    // kernel times the events bracket, profiles/r04_ab_results.txt block 18; default events kept)
    hipEvent_t e = nullptr;
    const hipError_t rc = hipEventCreate(&e);
    if (rc != hipSuccess) {
        fail(std::string("hipEventCreate: ") + hipGetErrorString(rc));
        return nullptr;
    }
    return e;
}

// take_sync_event() for the length of a scope: the event goes back to its net's pool on EVERY path out (HIP_TRY returns
// from the middle of a function; a wait has captured the recorded state by then, or nothing was recorded at all)
struct SyncEventScope {
    uva_net* n;
    hipEvent_t e;
    explicit SyncEventScope(uva_net* net) : n(net), e(take_sync_event(net)) {}
    ~SyncEventScope() { if (e) n->ev_sync_free.push_back(e); }
    SyncEventScope(const SyncEventScope&) = delete;
    SyncEventScope& operator=(const SyncEventScope&) = delete;
};

// an event that orders one stream behind another (default flags: its release makes the producer's writes visible)
hipEvent_t take_sync_event(uva_net* n)
{
    if (!n->ev_sync_free.empty()) {
        hipEvent_t e = n->ev_sync_free.back();
        n->ev_sync_free.pop_back();
        return e;
    }
    hipEvent_t e = nullptr;
    const hipError_t rc = hipEventCreateWithFlags(&e, hipEventDisableTiming);
    if (rc != hipSuccess) {
        fail(std::string("hipEventCreate: ") + hipGetErrorString(rc));
        return nullptr;
    }
    return e;
}

// Frames whose last event has completed are added to the statistics and their events recycled; frames still
// in flight (pipelined submits) stay pending for the next call.
void resolve_events(uva_net* n)
{
    std::vector<uva_net::EvSet> keep;
    for (auto& s : n->ev_pending) {
        if (hipEventQuery(s.e[3]) == hipErrorNotReady) {
            keep.push_back(s);
            continue;
        }
        float ms[3] = {0, 0, 0};
        bool ok = true;
        for (int k = 0; k < 3; ++k) ok &= hipEventElapsedTime(&ms[k], s.e[k], s.e[k + 1]) == hipSuccess;
        if (ok) {
            n->launches[0] += 1; n->total_ms[0] += ms[0];
            n->launches[1] += s.ntrunk; n->total_ms[1] += ms[1];
            n->launches[2] += 1; n->total_ms[2] += ms[2];
        }
        for (auto e : s.e) n->ev_free.push_back(e);
    }
    n->ev_pending.swap(keep);
    std::vector<uva_net::EvPair> keep2;
    for (auto& q : n->ev_pairs) {
        if (hipEventQuery(q.b) == hipErrorNotReady) { keep2.push_back(q); continue; }
        float ms = 0;
        if (hipEventElapsedTime(&ms, q.a, q.b) == hipSuccess) { n->launches[q.kind] += 1; n->total_ms[q.kind] += ms; }
        n->ev_free.push_back(q.a);
        n->ev_free.push_back(q.b);
    }
    n->ev_pairs.swap(keep2);
}

// The whole graph for one frame.  stop_after >= 0: run only convolutions 0..stop_after (debug).
// does this call go through sub10_kernel / sub5_kernel (the 24-feature 1x net on a whole-frame plane, u8 in / u8 out)?
bool sub10_route(const uva_net* n, const Workspace* ws, bool f32, int stop_after)
{
    const Graph& g = n->g;
    return n->fuse_all && !f32 && stop_after < 0 && g.nf == 24 && g.scale == 1 && (int)g.convs.size() == S10_NL && ws->planes.size() == 1 &&
           n->layers[0].wpk_s10 && !ws->sub10_unfit;
}

// `frames` frames (1 .. S10_MAXB) of the workspace's geometry through the 1x net in ONE launch, with the profiling events of one
// launch around it; 0 = done, 1 = error, 2 = does not fit (fewer frames, or for one frame the layer-pair path)
int run_sub10(uva_net* n, Workspace* ws, const void* const* srcs, size_t src_stride, void* const* dsts, size_t dst_stride, int frames)
{
    uva_net::EvSet ev1;
    const bool prof1 = n->prof;
    if (prof1) {
        for (auto& e : ev1.e) e = nullptr;
        for (auto& e : ev1.e) {
            e = take_event(n);
            if (!e) return 1;
        }
        HIP_TRY(hipEventRecord(ev1.e[0], n->stream));
        HIP_TRY(hipEventRecord(ev1.e[1], n->stream));
    }
    int rc = 2, launches = 2;
    if (frames == 1 && n->split5 && !ws->sub5_unfit) rc = launch_sub5(n, ws, srcs[0], src_stride, dsts[0], dst_stride);
    if (rc == 2) { rc = launch_sub10(n, ws, srcs, src_stride, dsts, dst_stride, frames); launches = 1; }
    if (prof1) {
        if (rc == 0) {
            HIP_TRY(hipEventRecord(ev1.e[2], n->stream));
            HIP_TRY(hipEventRecord(ev1.e[3], n->stream));
            ev1.ntrunk = launches;
            n->ev_pending.push_back(ev1);
        } else {
            for (auto e : ev1.e) n->ev_free.push_back(e);
        }
    }
    return rc;
}

int run_graph(uva_net* n, Workspace* ws, bool f32, const void* src, size_t src_stride, void* dst,
              size_t dst_stride, int stop_after)
{
    const Graph& g = n->g;
    const int nconv = (int)g.convs.size();
    // the 24-feature 1x net on a whole-frame plane, u8 in / u8 out: one launch for all ten convolutions
    if (sub10_route(n, ws, f32, stop_after)) {
        const int rc = run_sub10(n, ws, &src, src_stride, &dst, dst_stride, 1);
        if (rc != 2) return rc;
    }
    const bool prof = n->prof && stop_after < 0;
    uva_net::EvSet ev;
    ev.ntrunk = 0;   // trunk launches of this frame (a fused pair is one launch)
    if (prof) {
        for (auto& e : ev.e) e = nullptr;
        for (auto& e : ev.e) {
            e = take_event(n);
            if (!e) {
                for (auto q : ev.e)
                    if (q) n->ev_free.push_back(q);
                return 1;
            }
        }
        HIP_TRY(hipEventRecord(ev.e[0], n->stream));
    }
    HeadArgs ha;
    std::memset(&ha, 0, sizeof ha);
    ha.planes = ws->d_planes;
    ha.nplanes = (int)ws->planes.size();
    ha.ntiles = ws->ntiles;
    ha.src_u8 = f32 ? nullptr : (const uint8_t*)src;
    ha.src_stride = src_stride;
    ha.src_f32 = f32 ? (const float*)src : nullptr;
    ha.out_act = ws->act[0];
    ha.wpk = n->layers[0].wpk;
    ha.bias = n->layers[0].bias;
    ha.slope = n->layers[0].slope;
    ha.in_scale = f32 ? 1.0f : (float)(1 / 255.0);
    ha.sink = n->d_sink;
    if (launch_head(n, f32, ha)) return 1;
    if (prof) HIP_TRY(hipEventRecord(ev.e[1], n->stream));

    ConvArgs ca;
    std::memset(&ca, 0, sizeof ca);
    ca.planes = ws->d_planes;
    ca.nplanes = (int)ws->planes.size();
    ca.ntiles = ws->ntiles;
    ca.tiles_per_xcd = (ws->ntiles + 7) / 8;
    ca.sink = n->d_sink;
    // ping-pong: every launch (one trunk layer, or a fused pair of them) reads act[cur] and writes act[cur ^ 1]
    int cur = 0;
    n->last_act_buf = 0;
    bool negated = false;         // the current activation image holds the last layer's slope > 1 channels negated (see below)
    for (int i = 1; i < nconv - 1; ++i) {
        if (stop_after >= 0 && i > stop_after) return 0;
        // two trunk layers per launch where a pair is wanted in full (a debug tap on layer i itself runs it alone)
        if (n->fuse_pairs && n->wino && ws->d_stepsw && n->layers[i].wpk_w && i + 1 < nconv - 1 && (stop_after < 0 || i + 1 <= stop_after)) {
            // The packed-fp16 PReLU computes channels with a slope above 1 negated (uva_wino.h).  Where the NEXT launch is a
            // trunkw pair as well, this launch leaves its second layer's that way in HBM (no sign restore: four instructions per
            // block row) and the next launch's first layer reads them through weights packed for it (wpk_wn).  Whole frames only:
            // a debug tap (stop_after) keeps every image in the plain convention.
            const bool next_is_pair = stop_after < 0 && i + 3 < nconv - 1 && n->layers[i + 2].wpk_wn != nullptr;
            const bool carry_out = n->act16 && n->carry && next_is_pair && n->layers[i + 1].flip_w;
            TrunkwArgs wa;
            std::memset(&wa, 0, sizeof wa);
            wa.in_act = ws->act_base[cur];
            wa.out_act = ws->act_base[cur ^ 1];
            wa.steps = ws->d_stepsw;
            wa.nsteps = ws->d_nstepsw;
            wa.max_steps = ws->max_stepsw;
            wa.sink = n->d_sink;
            if (launch_trunkw(n, ws, wa, i, negated, carry_out)) return 1;
            negated = carry_out;
            ++ev.ntrunk;
            ++i;
            cur ^= 1;
            n->last_act_buf = cur;
            continue;
        }
        if (n->fuse_pairs && ws->d_steps2 && i + 1 < nconv - 1 && (stop_after < 0 || i + 1 <= stop_after)) {
            Trunk2Args ta;
            std::memset(&ta, 0, sizeof ta);
            ta.in_act = ws->act_base[cur];
            ta.out_act = ws->act_base[cur ^ 1];
            for (int k = 0; k < 2; ++k) {
                ta.wpk[k] = n->layers[i + k].wpk;
                ta.bias[k] = n->layers[i + k].bias;
                ta.slope[k] = n->layers[i + k].slope;
            }
            ta.steps = ws->d_steps2;
            ta.nsteps = ws->d_nsteps2;
            ta.max_steps = ws->max_steps2;
            ta.sink = n->d_sink;
            if (launch_trunk2(n, ws, ta)) return 1;
            ++ev.ntrunk;
            ++i;
            cur ^= 1;
            n->last_act_buf = cur;
            continue;
        }
        ca.in_act = ws->act[cur];
        ca.out_act = ws->act[cur ^ 1];
        ca.wpk = n->layers[i].wpk;
        ca.bias = n->layers[i].bias;
        ca.slope = n->layers[i].slope;
        ca.reverse = i & 1;   // layer 1 walks backwards over what the head wrote last, layer 2 forwards, ...
        if (n->fuse_pairs && g.nf == 24 && i + 1 < nconv - 1 && (stop_after < 0 || i + 1 <= stop_after)) {
            ca.wpk2 = n->layers[i + 1].wpk;
            ca.bias2 = n->layers[i + 1].bias;
            ca.slope2 = n->layers[i + 1].slope;
            ca.ntiles = ws->ntiles;
            ca.tiles_per_xcd = (ws->ntiles + 7) / 8;
            if (launch_pair24(n, ca)) return 1;
            ++ev.ntrunk;
            ++i;
            cur ^= 1;
            n->last_act_buf = cur;
            continue;
        }
        if (launch_trunk(n, ws, ca)) return 1;
        ++ev.ntrunk;
        cur ^= 1;
        n->last_act_buf = cur;
    }
    if (stop_after >= 0) return 0;
    if (prof) HIP_TRY(hipEventRecord(ev.e[2], n->stream));
    ca.ntiles = ws->ntiles;
    ca.tiles_per_xcd = (ws->ntiles + 7) / 8;
    ca.in_act = ws->act[cur];
    ca.reverse = (nconv - 1) & 1;
    ca.out_act = nullptr;
    ca.wpk = n->layers[nconv - 1].wpk;
    ca.bias = n->layers[nconv - 1].bias;
    ca.slope = nullptr;
    if (f32) {
        ca.src_f32 = (const float*)src;
        ca.dst_f32 = (float*)dst;
    } else {
        ca.src_u8 = (const uint8_t*)src;
        ca.src_stride = src_stride;
        ca.dst_u8 = (uint8_t*)dst;
        ca.dst_stride = dst_stride;
    }
    if (f32 ? launch_conv(n, 2, ca) : launch_tail_u8(n, ws, ca)) return 1;
    if (prof) {
        HIP_TRY(hipEventRecord(ev.e[3], n->stream));
        n->ev_pending.push_back(ev);
    }
    return 0;
}

template <typename T>
int grow_dev(T** p, size_t* cap, size_t bytes)
{
    if (*cap >= bytes) return 0;
    if (*p) HIP_TRY(hipFree(*p));
    *p = nullptr; *cap = 0;
    HIP_TRY(hipMalloc((void**)p, bytes));
    *cap = bytes;
    return 0;
}

int grow_host(uint8_t** p, size_t* cap, size_t bytes)
{
    if (*cap >= bytes) return 0;
    if (*p) HIP_TRY(hipHostFree(*p));
    *p = nullptr; *cap = 0;
    HIP_TRY(hipHostMalloc((void**)p, bytes, hipHostMallocDefault));
    *cap = bytes;
    return 0;
}

// ---- generic graphs -------------------------------------------------------------------------------
int generic_acquire(uva_net* n, int h, int w, int c, GBuf* out)
{
    out->h = h; out->w = w; out->c = c; out->cpad = GenericDevice::pad32(c);
    auto& free_list = n->gd.pool[std::make_tuple(h, w, c)];
    if (!free_list.empty()) {
        out->p = free_list.back();
        free_list.pop_back();
        return 0;
    }
    const size_t bytes = out->elems() * 2;
    HIP_TRY(hipMalloc((void**)&out->p, bytes));
    HIP_TRY(hipMemsetAsync(out->p, 0, bytes, n->stream));    // border and padding channels stay zero for the array's life
    n->gd.pool_bytes += bytes;
    return 0;
}

void generic_release(uva_net* n, const GBuf& b)
{
    if (b.p) n->gd.pool[std::make_tuple(b.h, b.w, b.c)].push_back(b.p);
}

// One BATCH of planes (reference tiles, or the whole frame) through the graph in lockstep: layer by layer, every plane's
// launch of that layer -- and for the residual-dense-block kernels (csrc/uva_rdb.hip.h) ONE launch for all planes, their
// arrays in a table: a frame's small planes then share the workgroups with the large ones instead of under-filling the
// chip in launches of their own.  Arrays are recycled as soon as their last reader has been queued (stream order makes
// that safe) and only ever for a blob of the same shape, so borders and padding channels -- never written -- stay zero.
// All planes of a batch take the same kernels (generic_plane_class: the same side of every width threshold).
struct PlaneJob {
    const void* src; size_t src_stride; int sy0, sx0, h, w;
    void* dst; size_t dst_stride; int cy0, cy1, cx0, cx1;
};
// workgroups of the persistent kernels: one per CU (UVA_GENERIC_GRID=K: a test hook -- few workgroups give long segments
// and several of them per workgroup on frames small enough for the numpy restatement)
inline int generic_grid(const uva_net* n)
{
    static const int forced = [] { const char* e = uva::debug_env("UVA_GENERIC_GRID"); return e ? std::atoi(e) : 0; }();
    return forced > 0 ? forced : std::max(8, (n->ncu / 8) * 8);
}
// (the strip kernels want 32 / 64 output columns at 1x, 2x or 4x the plane's width; a batch is planned with its narrowest
// plane, so mixing classes would only cost speed, never change a result)
inline int generic_plane_class(int w) { return (w >= 8) + (w >= 16) + (w >= 32) + (w >= 64); }

int generic_run_planes(uva_net* n, bool f32, const std::vector<PlaneJob>& jobs)
{
    const GenericGraph& g = n->gg;
    const int np = (int)jobs.size();
    if (np <= 0 || np > GEN_MAX_PLANES) return fail("generic executor: bad plane batch");
    auto root = [&](int b) { while (g.blobs[b].alias_of >= 0) b = g.blobs[b].alias_of; return b; };
    // dense chains (uva_generic.h plan_concat_groups): the blobs of a chain are channel ranges of one shared array, which
    // goes back to the pool when the last of them has been read.  Needs g_conv3_lds (prefix reads, strided writes).
    const bool use_groups = n->generic_lds_conv;
    auto group_of = [&](int b) { return use_groups ? g.blobs[b].group : -1; };
    struct PlaneState {
        PlaneJob j;
        std::vector<GBuf> buf, garr;
        std::vector<int> left, glive;
    };
    std::vector<PlaneState> P(np);
    int wmin = jobs[0].w;
    for (int pi = 0; pi < np; ++pi) {
        PlaneState& ps = P[pi];
        ps.j = jobs[pi];
        ps.buf.assign(g.blobs.size(), GBuf());
        ps.left.assign(g.blobs.size(), 0);
        for (size_t b = 0; b < g.blobs.size(); ++b) ps.left[b] = g.blobs[b].consumers;
        ps.garr.assign(g.group_channels.size(), GBuf());
        ps.glive.assign(g.group_blobs.begin(), g.group_blobs.end());
        wmin = std::min(wmin, jobs[pi].w);
    }
    auto done_with = [&](PlaneState& ps, int b) {
        b = root(b);
        if (--ps.left[b] != 0) return;
        const int gi = group_of(b);
        if (gi >= 0) {
            ps.buf[b].p = nullptr;
            if (--ps.glive[gi] == 0) { generic_release(n, ps.garr[gi]); ps.garr[gi].p = nullptr; }
        } else {
            generic_release(n, ps.buf[b]);
            ps.buf[b].p = nullptr;
        }
    };
    struct Cleanup {
        uva_net* n; std::vector<PlaneState>* P; const GenericGraph* g; bool use_groups;
        ~Cleanup()
        {
            for (PlaneState& ps : *P) {
                for (size_t b = 0; b < ps.buf.size(); ++b)
                    if (ps.buf[b].p && !(use_groups && g->blobs[b].group >= 0)) generic_release(n, ps.buf[b]);
                for (auto& a : ps.garr) if (a.p) generic_release(n, a);
            }
        }
    } cleanup{n, &P, &g, use_groups};
    const int T = 256;
    // Element-wise sums that directly follow a convolution of g_conv3_lds are done in its epilogue (GConvArgs::res): the
    // convolution's own result never goes to memory, the sum layer is skipped.  Conditions: the convolution's result has
    // no other reader, the other operand exists when the convolution runs, whole 16-byte channel units.
    std::vector<int> fuse_add(g.layers.size(), -1), fuse_pos(g.layers.size(), 0);
    std::vector<char> skip(g.layers.size(), 0);
    if (n->generic_lds_conv && n->generic_fuse_add) {
        std::vector<int> producer(g.blobs.size(), -1);
        for (size_t li = 0; li < g.layers.size(); ++li)
            if (g.layers[li].kind != GLayer::SPLIT && !g.layers[li].out.empty()) producer[g.layers[li].out[0]] = (int)li;
        for (size_t ri = 0; ri < g.layers.size(); ++ri) {
            const GLayer& r = g.layers[ri];
            if ((r.kind != GLayer::ADD && r.kind != GLayer::ELTWISE_SUM) || r.in.size() != 2 || r.coeffs.size() != 2) continue;
            for (int k = 0; k < 2 && !skip[ri]; ++k) {
                const int cb = root(r.in[k]), ob2 = root(r.in[1 - k]);
                const int li = producer[cb];
                if (li < 0 || cb == ob2 || g.layers[li].kind != GLayer::CONV || fuse_add[li] >= 0) continue;
                if (g.blobs[cb].consumers != 1 || g.blobs[cb].channels % 8 || group_of(cb) >= 0) continue;
                if (!n->gd.convs[g.layers[li].conv].wpk_lds || producer[ob2] < 0 || producer[ob2] >= li) continue;
                fuse_add[li] = (int)ri;
                fuse_pos[li] = k;
                skip[ri] = 1;
            }
        }
    }
    // g_conv3_sw (the 192 -> 64 and 64 -> 64 convolutions on planes wide enough for its strips) also takes a SECOND sum
    // behind the first -- the `rrdb_in*1.0 + rdb_out*0.2` that closes every third dense block of 4x_Valar_v1
    // (models/4x_Valar_v1.param:55-56): the first sum's result then never goes to memory either.
    static const bool sw_on = [] { const char* e = uva::debug_env("UVA_GENERIC_SW"); return !e || std::atoi(e) != 0; }();
    auto sw_cols_of = [&](size_t li) -> int {      // strip width g_conv3_sw would use for layer li on this plane, 0: not its case
        const GLayer& l = g.layers[li];
        if (!sw_on || !n->generic_lds_conv || l.kind != GLayer::CONV || l.ksize != 3) return 0;
        const GenericDevice::ConvDev& cd = n->gd.convs[l.conv];
        if (cd.cout_pad != 64 || g.blobs[l.out[0]].channels != 64 || (cd.cin_pad != 64 && cd.cin_pad != 192)) return 0;
        const int cols = cd.cin_pad == 192 ? sw_cols<1>() : sw_cols<2>();
        return wmin * g.blobs[l.out[0]].scale >= cols ? cols : 0;
    };
    // the instantiations of g_conv3_sw that exist (what 4x_Valar_v1 needs): -1 = none, the layer-by-layer kernel takes it
    auto sw_variant = [](int cin_pad, bool act, int rm, int rm2) -> int {
        if (cin_pad == 192 && !act && rm == 2 && rm2 == 0) return 0;
        if (cin_pad == 192 && !act && rm == 2 && rm2 == 2) return 1;
        if (cin_pad == 64 && !act && rm == 1 && rm2 == 0) return 2;
        if (cin_pad == 64 && act && rm == 0 && rm2 == 0) return 3;
        if (cin_pad == 192 && !act && rm == 0 && rm2 == 0) return 4;     // (UVA_GENERIC_FUSE_ADD=0: the sums as launches of their own)
        if (cin_pad == 64 && !act && rm == 0 && rm2 == 0) return 5;
        return -1;
    };
    std::vector<int> fuse_add2(g.layers.size(), -1), fuse_pos2(g.layers.size(), 0);
    static const bool add2_on = [] { const char* e = uva::debug_env("UVA_GENERIC_FUSE_ADD2"); return !e || std::atoi(e) != 0; }();
    if (n->generic_lds_conv && n->generic_fuse_add && add2_on) {
        std::vector<int> producer(g.blobs.size(), -1);
        std::vector<std::vector<int>> readers(g.blobs.size());
        for (size_t li = 0; li < g.layers.size(); ++li) {
            if (g.layers[li].kind == GLayer::SPLIT) continue;
            if (!g.layers[li].out.empty()) producer[g.layers[li].out[0]] = (int)li;
            for (int b : g.layers[li].in) readers[root(b)].push_back((int)li);
        }
        for (size_t li = 0; li < g.layers.size(); ++li) {
            if (fuse_add[li] < 0 || !sw_cols_of(li)) continue;
            const int o1 = root(g.layers[fuse_add[li]].out[0]);
            if (g.blobs[o1].consumers != 1 || readers[o1].size() != 1 || group_of(o1) >= 0 || o1 == root(g.out_blob)) continue;
            const int ri = readers[o1][0];
            const GLayer& r = g.layers[ri];
            if ((r.kind != GLayer::ADD && r.kind != GLayer::ELTWISE_SUM) || r.in.size() != 2 || r.coeffs.size() != 2 || skip[ri]) continue;
            const int k = root(r.in[0]) == o1 ? 0 : 1, ob3 = root(r.in[1 - k]);
            if (ob3 == o1 || producer[ob3] < 0 || producer[ob3] >= (int)li) continue;
            const GLayer& cl = g.layers[li];
            if (sw_variant(n->gd.convs[cl.conv].cin_pad, cl.has_act, fuse_pos[li] == 1 ? 1 : 2, k == 1 ? 1 : 2) < 0) continue;
            fuse_add2[li] = ri;
            fuse_pos2[li] = k;
            skip[ri] = 1;
        }
    }
    // A nearest-neighbour 2x Interp whose only reader is a 64 -> 64 convolution of g_conv3_sw is folded into that
    // convolution's row DMA (g_conv3_sw<..., UP>): the enlarged array is never written (models/4x_Valar_v1.param:1000-1003:
    // at 4x it is 16x the 1x plane).  UVA_GENERIC_FUSE_INTERP=0: the Interp as a launch of its own, the A/B switch.
    static const bool up_on = [] { const char* e = uva::debug_env("UVA_GENERIC_FUSE_INTERP"); return !e || std::atoi(e) != 0; }();
    std::vector<int> up_of(g.layers.size(), -1);
    if (up_on) {
        std::vector<int> producer(g.blobs.size(), -1);
        for (size_t li = 0; li < g.layers.size(); ++li)
            if (g.layers[li].kind != GLayer::SPLIT && !g.layers[li].out.empty()) producer[g.layers[li].out[0]] = (int)li;
        for (size_t li = 0; li < g.layers.size(); ++li) {
            const GLayer& cl = g.layers[li];
            if (!sw_cols_of(li) || fuse_add[li] >= 0 || cl.in.size() != 1) continue;
            if (sw_variant(n->gd.convs[cl.conv].cin_pad, cl.has_act, 0, 0) != 3) continue;
            const int ib = root(cl.in[0]), pi = producer[ib];
            if (pi < 0 || g.layers[pi].kind != GLayer::INTERP_NEAREST || g.layers[pi].factor != 2 || g.blobs[ib].consumers != 1) continue;
            if (group_of(ib) >= 0 || group_of(root(g.layers[pi].in[0])) >= 0 || ib == root(g.out_blob)) continue;
            up_of[li] = pi;
            skip[pi] = 1;
        }
    }
    // On the u8 route the graph's last convolution (3 output channels, no activation, no sum, g_conv3_lds) writes the frame's
    // bytes itself (GConvArgs::u8dst): no fp16 result array, no g_output_u8 launch.  UVA_GENERIC_FUSE_OUT=0: the A/B switch.
    static const bool out_on = [] { const char* e = uva::debug_env("UVA_GENERIC_FUSE_OUT"); return !e || std::atoi(e) != 0; }();
    int out_conv = -1;
    if (out_on && !f32 && n->generic_lds_conv) {
        for (size_t li = 0; li < g.layers.size(); ++li) {
            const GLayer& l = g.layers[li];
            if (l.kind != GLayer::CONV || l.out.empty() || root(l.out[0]) != root(g.out_blob)) continue;
            const GenericDevice::ConvDev& cd = n->gd.convs[l.conv];
            const GBlob& ob = g.blobs[l.out[0]];
            if (cd.wpk_lds && l.ksize == 3 && !l.has_act && fuse_add[li] < 0 && ob.channels == 3 && group_of(l.out[0]) < 0 && ob.scale == g.scale)
                out_conv = (int)li;
        }
    }
    // residual dense blocks whose first four convolutions run as one rdb4_kernel launch at the first one (UVA_GENERIC_RDB=0:
    // layer by layer, the A/B switch): the other six layers are bookkeeping only, and none of their sums is fused elsewhere
    static const bool rdb_on = [] { const char* e = uva::debug_env("UVA_GENERIC_RDB"); return !e || std::atoi(e) != 0; }();
    std::vector<int> rdb_at(g.layers.size(), -1);
    std::vector<char> rdb_skip(g.layers.size(), 0);
    if (rdb_on && use_groups && wmin >= 16) {
        for (size_t k = 0; k < n->gd.rdbs.size(); ++k) {
            const RdbMatch& m = n->gd.rdbs[k];
            rdb_at[m.c1] = (int)k;
            for (int li : {m.c2, m.c2s, m.add2, m.c3, m.c4, m.add4}) {
                rdb_skip[li] = 1;
                skip[li] = 0;
                fuse_add[li] = -1;
            }
        }
    }
    auto acquire_out = [&](PlaneState& ps, const GLayer& ly, GBuf* out) -> int {
        const GBlob& ob = g.blobs[ly.out[0]];
        GBuf o;
        if (group_of(ly.out[0]) >= 0) {
            GBuf& ga = ps.garr[ob.group];
            if (!ga.p && generic_acquire(n, ps.j.h * ob.scale, ps.j.w * ob.scale, g.group_channels[ob.group], &ga)) return 1;
            o = ga;                                   // same geometry and pixel stride (cpad) ...
            o.p = ga.p + ob.group_off;                // ... starting at the blob's first channel
            o.c = ob.channels;
        } else if (generic_acquire(n, ps.j.h * ob.scale, ps.j.w * ob.scale, ob.channels, &o)) return 1;
        ps.buf[ly.out[0]] = o;
        *out = o;
        return 0;
    };
    for (size_t layer_i = 0; layer_i < g.layers.size(); ++layer_i) {
        const GLayer& gl = g.layers[layer_i];
        if (gl.kind == GLayer::SPLIT || skip[layer_i]) continue;
        if (rdb_skip[layer_i]) {          // done by the rdb4 launch at the block's first convolution: buffers and counts only
            for (PlaneState& ps : P) {
                GBuf o;
                if (group_of(gl.out[0]) >= 0 && acquire_out(ps, gl, &o)) return 1;
                for (int b : gl.in) done_with(ps, b);
            }
            continue;
        }
        const GLayer* const sum = fuse_add[layer_i] >= 0 ? &g.layers[fuse_add[layer_i]] : nullptr;
        const GLayer* const sum2 = fuse_add2[layer_i] >= 0 ? &g.layers[fuse_add2[layer_i]] : nullptr;
        const GLayer* const up = gl.kind == GLayer::CONV && up_of[layer_i] >= 0 ? &g.layers[up_of[layer_i]] : nullptr;
        auto finish_layer = [&](PlaneState& ps) {
            if (up) done_with(ps, up->in[0]);          // (the Interp's own result was never made)
            else for (int b : gl.in) done_with(ps, b);
            if (sum) done_with(ps, sum->in[1 - fuse_pos[layer_i]]);
            if (sum2) done_with(ps, sum2->in[1 - fuse_pos2[layer_i]]);
        };
        // ---- the kernels that take all planes in one launch ----------------------------------------------------------
        if (gl.kind == GLayer::CONV && rdb_at[layer_i] >= 0) {
            const RdbMatch& m = n->gd.rdbs[rdb_at[layer_i]];
            auto cdev = [&](int li) -> const GenericDevice::ConvDev& { return n->gd.convs[g.layers[li].conv]; };
            RdbArgs ra;
            std::memset(&ra, 0, sizeof ra);
            std::vector<int> dims;
            for (int pi = 0; pi < np; ++pi) {
                GBuf o;
                if (acquire_out(P[pi], gl, &o)) return 1;
                const GBuf& arr = P[pi].garr[m.group];
                ra.arr[pi] = arr.p; ra.ph[pi] = arr.h; ra.pw[pi] = arr.w; ra.stride = arr.cpad;
                dims.push_back(arr.h); dims.push_back(arr.w);
            }
            GenericDevice::RdbPlan& plan = n->gd.rdb_plans[dims];
            if (!plan.segs) {
                std::vector<RdbSeg> segs;
                std::vector<int> sbeg;
                plan.grid = generic_grid(n);
                rdb_segments(dims, plan.grid, segs, sbeg);
                if (upload(&plan.segs, segs.data(), segs.size() * sizeof(RdbSeg), n->stream)) return 1;
                if (upload(&plan.seg_begin, sbeg.data(), sbeg.size() * sizeof(int), n->stream)) return 1;
                HIP_TRY(hipStreamSynchronize(n->stream));
            }
            ra.w1 = cdev(m.c1).wpk; ra.w2 = cdev(m.c2).wpk; ra.w2s = cdev(m.c2s).wpk; ra.w3 = cdev(m.c3).wpk; ra.w4 = cdev(m.c4).wpk;
            ra.b1 = cdev(m.c1).bias; ra.b2 = cdev(m.c2).bias; ra.b3 = cdev(m.c3).bias; ra.b4 = cdev(m.c4).bias;
            ra.slope = m.slope;
            ra.segs = plan.segs; ra.seg_begin = plan.seg_begin; ra.sink = n->d_sink;
#ifdef UVA_INSTRUMENT
            if (uva::debug_env("UVA_RDB_STAMPS")) {
                if (!n->gd.rdb_dbg) HIP_TRY(hipMalloc((void**)&n->gd.rdb_dbg, 1024 * 16 * 8));
                HIP_TRY(hipMemsetAsync(n->gd.rdb_dbg, 0, 1024 * 16 * 8, n->stream));
                ra.dbg = n->gd.rdb_dbg;
            }
#endif
            if (!n->attr_set[28]) {
                HIP_TRY(hipFuncSetAttribute((const void*)rdb4_kernel, hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024));
                n->attr_set[28] = true;
            }
            EvPairScope evp(n, 1, n->prof);
            HIP_TRY(evp.begin());
            hipLaunchKernelGGL(rdb4_kernel, dim3(plan.grid), dim3(256), rdb4_lds_bytes(), n->stream, ra);
            HIP_TRY(hipGetLastError());
            HIP_TRY(evp.end());
            for (PlaneState& ps : P) finish_layer(ps);
            continue;
        }
        if (gl.kind == GLayer::CONV && sw_cols_of(layer_i)) {
            const GenericDevice::ConvDev& cd = n->gd.convs[gl.conv];
            const int rm = !sum ? 0 : fuse_pos[layer_i] == 1 ? 1 : 2, rm2 = !sum2 ? 0 : fuse_pos2[layer_i] == 1 ? 1 : 2;
            const int variant = sw_variant(cd.cin_pad, gl.has_act, rm, rm2);
            if (variant >= 0) {
                const int cols = sw_cols_of(layer_i);
                GSwArgs sa;
                std::memset(&sa, 0, sizeof sa);
                std::vector<int> dims{cols};
                for (int pi = 0; pi < np; ++pi) {
                    PlaneState& ps = P[pi];
                    GBuf o;
                    if (acquire_out(ps, sum2 ? *sum2 : sum ? *sum : gl, &o)) return 1;
                    const GBuf& a = ps.buf[root(up ? up->in[0] : gl.in[0])];
                    sa.in[pi] = a.p; sa.out[pi] = o.p; sa.ph[pi] = o.h; sa.pw[pi] = o.w;
                    sa.in_stride = a.cpad; sa.out_stride = o.cpad;
                    if (sum) {
                        const GBuf& other = ps.buf[root(sum->in[1 - fuse_pos[layer_i]])];
                        sa.res[pi] = other.p; sa.res_stride = other.cpad;
                    }
                    if (sum2) {
                        const GBuf& other = ps.buf[root(sum2->in[1 - fuse_pos2[layer_i]])];
                        sa.res2[pi] = other.p; sa.res2_stride = other.cpad;
                    }
                    dims.push_back(o.h); dims.push_back(o.w);
                }
                GenericDevice::SwPlan& plan = n->gd.sw_plans[dims];
                if (!plan.segs) {
                    std::vector<GSwSeg> segs;
                    std::vector<int> sbeg;
                    plan.grid = generic_grid(n);
                    sw_segments(std::vector<int>(dims.begin() + 1, dims.end()), cols, plan.grid, segs, sbeg);
                    if (upload(&plan.segs, segs.data(), segs.size() * sizeof(GSwSeg), n->stream)) return 1;
                    if (upload(&plan.seg_begin, sbeg.data(), sbeg.size() * sizeof(int), n->stream)) return 1;
                    HIP_TRY(hipStreamSynchronize(n->stream));       // (the vectors go away)
                }
                sa.wpk = cd.wpk; sa.bias = cd.bias; sa.out_coff = 0; sa.slope = gl.act_slope;
                if (sum) { sa.ca = sum->coeffs[0]; sa.cb = sum->coeffs[1]; }
                if (sum2) { sa.ca2 = sum2->coeffs[0]; sa.cb2 = sum2->coeffs[1]; }
                sa.segs = plan.segs; sa.seg_begin = plan.seg_begin; sa.sink = n->d_sink;
                auto launch_sw = [&](auto kern, int slot, size_t lds) -> int {
                    if (!n->attr_set[slot]) {
                        HIP_TRY(hipFuncSetAttribute((const void*)kern, hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024));
                        n->attr_set[slot] = true;
                    }
                    EvPairScope evp(n, 2, n->prof && cd.cin_pad == 192);
                    HIP_TRY(evp.begin());
                    hipLaunchKernelGGL(kern, dim3(plan.grid), dim3(256), lds, n->stream, sa);
                    HIP_TRY(evp.end());
                    return 0;
                };
                // 192 inputs: UVA_GENERIC_SK=1 takes g_conv3_sk (32x32x16 MFMAs, the k-loop split between wave pairs) instead of
                // g_conv3_sw<6, 1>.  Measured equal within 1 % (profiles/r03_ab_results.txt, block 10: both sit at the package's
                // power limit), so the simpler kernel stays the default; the other is kept runnable for the next round's work.
                static const bool sk_on = [] { const char* e = uva::debug_env("UVA_GENERIC_SK"); return e && std::atoi(e) != 0; }();
                // 192 inputs, round 6: g_conv3_sww -- the same convolution as 1-D Winograd F(2,3), two thirds of the MFMAs
                // (csrc/uva_sww.hip.h).  UVA_GENERIC_WINO=0: the direct kernels below, the A/B switch.
                static const bool wino_on = [] { const char* e = uva::debug_env("UVA_GENERIC_WINO"); return !e || std::atoi(e) != 0; }();
                if ((variant == 0 || variant == 1 || variant == 4) && wino_on && !sk_on && cd.wpk_w) {
                    sa.wpk = cd.wpk_w;
                    EvPairScope evp(n, 2, n->prof);
                    HIP_TRY(evp.begin());
                    HIP_TRY(launch_conv3_sww(n->stream, plan.grid, sa, variant == 4 ? 0 : 2, variant == 1 ? 2 : 0));
                    HIP_TRY(evp.end());
                }
                else if (variant == 0 && sk_on) { if (launch_sw(g_conv3_sk<2, 0>, 32, sk_lds_bytes())) return 1; }
                else if (variant == 1 && sk_on) { if (launch_sw(g_conv3_sk<2, 2>, 33, sk_lds_bytes())) return 1; }
                else if (variant == 4 && sk_on) { if (launch_sw(g_conv3_sk<0, 0>, 34, sk_lds_bytes())) return 1; }
                else if (variant == 0) { if (launch_sw(g_conv3_sw<6, 1, false, 2, 0>, 24, sw_lds_bytes<6, 1>())) return 1; }
                else if (variant == 1) { if (launch_sw(g_conv3_sw<6, 1, false, 2, 2>, 25, sw_lds_bytes<6, 1>())) return 1; }
                else if (variant == 2) { if (launch_sw(g_conv3_sw<2, 2, false, 1, 0>, 26, sw_lds_bytes<2, 2>())) return 1; }
                else if (variant == 3 && up) { if (launch_sw(g_conv3_sw<2, 2, true, 0, 0, true>, 31, sw_lds_bytes<2, 2>())) return 1; }
                else if (variant == 3) { if (launch_sw(g_conv3_sw<2, 2, true, 0, 0>, 27, sw_lds_bytes<2, 2>())) return 1; }
                else if (variant == 4) { if (launch_sw(g_conv3_sw<6, 1, false, 0, 0>, 29, sw_lds_bytes<6, 1>())) return 1; }
                else { if (launch_sw(g_conv3_sw<2, 2, false, 0, 0>, 30, sw_lds_bytes<2, 2>())) return 1; }
                HIP_TRY(hipGetLastError());
                for (PlaneState& ps : P) finish_layer(ps);
                continue;
            }
        }
        // ---- everything else: one launch per plane ------------------------------------------------------------------------
        for (PlaneState& ps : P) {
        std::vector<GBuf>& buf = ps.buf;
        const int h = ps.j.h, w = ps.j.w;
        GBuf o;
        if (acquire_out(ps, sum2 ? *sum2 : sum ? *sum : gl, &o)) return 1;       // (a fused convolution writes the last sum's array, it has none of its own)
        auto in = [&](int k) -> const GBuf& { return buf[root(gl.in[k])]; };
        switch (gl.kind) {
        case GLayer::INPUT:
            if (f32) hipLaunchKernelGGL(g_input_f32, dim3((w + T - 1) / T, h), dim3(T), 0, n->stream, (const float*)ps.j.src, h, w, o.p, o.cpad);
            else hipLaunchKernelGGL(g_input_u8, dim3((w + T - 1) / T, h), dim3(T), 0, n->stream, (const uint8_t*)ps.j.src, ps.j.src_stride, ps.j.sy0, ps.j.sx0, h, w, o.p, o.cpad);
            break;
        case GLayer::CONV: {
            const GenericDevice::ConvDev& cd = n->gd.convs[gl.conv];
            const GBuf& a = in(0);
            if ((group_of(gl.out[0]) >= 0 || group_of(root(gl.in[0])) >= 0) && !cd.wpk_lds)
                return fail("generic executor: a dense-chain convolution without the LDS kernel (plan_concat_groups and ensure_device disagree)");
            if (cd.wpk_lds && n->generic_lds_conv) {
                GConvArgs ga;
                std::memset(&ga, 0, sizeof ga);
                ga.in = a.p; ga.in_stride = a.cpad; ga.cin_pad = cd.cin_pad;
                ga.wpk = cd.wpk_lds; ga.bias = cd.bias;
                ga.out = o.p; ga.out_stride = o.cpad; ga.out_coff = 0; ga.cout = o.c;
                ga.h = a.h; ga.w = a.w;
                ga.has_act = gl.has_act ? 1 : 0; ga.slope = gl.act_slope;
                if (sum) {
                    const GBuf& other = buf[root(sum->in[1 - fuse_pos[layer_i]])];
                    ga.res = other.p; ga.res_stride = other.cpad; ga.res_first = fuse_pos[layer_i] == 1;
                    ga.ca = sum->coeffs[0]; ga.cb = sum->coeffs[1];
                }
                if ((int)layer_i == out_conv) {       // the frame's bytes leave from this convolution's epilogue (no g_output_u8)
                    const int s = g.scale;
                    ga.u8dst = (uint8_t*)ps.j.dst; ga.u8stride = ps.j.dst_stride;
                    ga.dy0 = ps.j.sy0 * s; ga.dx0 = ps.j.sx0 * s;
                    ga.cy0 = ps.j.cy0 * s; ga.cy1 = ps.j.cy1 * s; ga.cx0 = ps.j.cx0 * s; ga.cx1 = ps.j.cx1 * s;
                }
                const int mbn = cd.cout_pad / 16;
                // (UVA_GENERIC_WG=0: the 3x3 convolutions stage their weights through LDS again -- the A/B switch)
                static const int wg_max = [] { const char* e = uva::debug_env("UVA_GENERIC_WG"); return e ? std::atoi(e) : 4; }();
                const bool wg = gl.ksize == 3 && mbn <= wg_max && (mbn == 2 || mbn == 4);
                // 16-row tiles (8 waves) where the plane is tall enough to keep every CU busy with them
                static const int nw_max = [] { const char* e = uva::debug_env("UVA_GENERIC_NW"); return e ? std::atoi(e) : 4; }();
                const int nw = (wg && nw_max >= 8 && (long long)((a.w + GC_TW - 1) / GC_TW) * ((a.h + 15) / 16) >= 4LL * n->ncu) ? 8 : 4;
                const size_t lds = g_conv3_lds_bytes(cd.cin_pad, mbn, gl.ksize, wg, nw);
                const dim3 g3((a.w + GC_TW - 1) / GC_TW, (a.h + 2 * nw - 1) / (2 * nw));
                auto launch = [&](auto kern, int slot) -> int {
                    if (!n->attr_set[slot]) {
                        HIP_TRY(hipFuncSetAttribute((const void*)kern, hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024));
                        n->attr_set[slot] = true;
                    }
                    hipLaunchKernelGGL(kern, g3, dim3(64 * nw), lds, n->stream, ga);
                    return 0;
                };
                if (gl.ksize == 1) {
                    if (mbn == 1) { if (launch(g_conv3_lds<1, 1>, 16)) return 1; }
                    else if (mbn == 2) { if (launch(g_conv3_lds<2, 1>, 17)) return 1; }
                    else if (mbn == 3) { if (launch(g_conv3_lds<3, 1>, 18)) return 1; }
                    else { if (launch(g_conv3_lds<4, 1>, 19)) return 1; }
                }
                else if (wg && mbn == 2 && nw == 8) { if (launch(g_conv3_lds<2, 3, true, 8>, 23)) return 1; }
                else if (wg && mbn == 4 && nw == 8) { if (launch(g_conv3_lds<4, 3, true, 8>, 15)) return 1; }
                else if (wg && mbn == 2) { if (launch(g_conv3_lds<2, 3, true>, 21)) return 1; }
                else if (wg && mbn == 4) { if (launch(g_conv3_lds<4, 3, true>, 22)) return 1; }
                else if (mbn == 1) { if (launch(g_conv3_lds<1>, 11)) return 1; }
                else if (mbn == 2) { if (launch(g_conv3_lds<2>, 12)) return 1; }
                else if (mbn == 3) { if (launch(g_conv3_lds<3>, 13)) return 1; }
                else { if (launch(g_conv3_lds<4>, 14)) return 1; }
                break;
            }
            const dim3 grid((a.w + 63) / 64, (a.h + 3) / 4, (cd.cout_pad + 63) / 64);
            if (gl.ksize == 3)
                hipLaunchKernelGGL(g_conv<3>, grid, dim3(256), 0, n->stream, a.p, a.cpad, cd.wpk, cd.bias, o.p, o.c, cd.cout_pad, o.cpad, a.h, a.w, gl.has_act ? 1 : 0, gl.act_slope);
            else
                hipLaunchKernelGGL(g_conv<1>, grid, dim3(256), 0, n->stream, a.p, a.cpad, cd.wpk, cd.bias, o.p, o.c, cd.cout_pad, o.cpad, a.h, a.w, gl.has_act ? 1 : 0, gl.act_slope);
            break;
        }
        case GLayer::ADD:
        case GLayer::ELTWISE_SUM: {
            if (group_of(root(gl.in[0])) >= 0 || group_of(root(gl.in[1])) >= 0 || group_of(gl.out[0]) >= 0) {
                // an operand or the result is a channel range of a wider array (dense chain): per-pixel strides
                if (o.c % 8) return fail("generic executor: strided add needs a multiple of 8 channels");
                const size_t npix = (size_t)(o.h + 3) * (o.w + 2), work = npix * (o.c / 8);
                hipLaunchKernelGGL(g_axpby_strided, dim3((unsigned)((work + T - 1) / T)), dim3(T), 0, n->stream, in(0).p, in(0).cpad,
                                   gl.coeffs[0], in(1).p, in(1).cpad, gl.coeffs[1], o.p, o.cpad, o.c / 8, npix);
                break;
            }
            const size_t n8 = o.elems() / 8;
            hipLaunchKernelGGL(g_axpby, dim3((unsigned)((n8 + T - 1) / T)), dim3(T), 0, n->stream, (const half8*)in(0).p, gl.coeffs[0],
                               (const half8*)in(1).p, gl.coeffs[1], (half8*)o.p, n8);
            break;
        }
        case GLayer::CONCAT: {
            const size_t npix = (size_t)(o.h + 3) * (o.w + 2);
            int c_off = 0;
            const int mode = use_groups ? gl.concat_mode : 0;
            if (mode == 2) break;                     // every input already sits in its channel range of the shared array
            for (size_t k = 0; k < (mode == 1 ? 1 : gl.in.size()); ++k) {
                const GBuf& a = in((int)k);
                const size_t work = npix * (a.c / 8);
                hipLaunchKernelGGL(g_concat_part, dim3((unsigned)((work + T - 1) / T)), dim3(T), 0, n->stream, a.p, a.cpad, a.c, o.p, o.cpad, c_off, npix);
                c_off += a.c;
            }
            break;
        }
        case GLayer::INTERP_NEAREST: {
            const GBuf& a = in(0);
            const size_t work = (size_t)o.h * o.w * (o.cpad / 8);
            hipLaunchKernelGGL(g_interp_nearest, dim3((unsigned)((work + T - 1) / T)), dim3(T), 0, n->stream, a.p, a.h, a.w, a.cpad, o.p, gl.factor);
            break;
        }
        case GLayer::PRELU: {
            const size_t npix = (size_t)(o.h + 3) * (o.w + 2), work = npix * o.c;
            hipLaunchKernelGGL(g_prelu, dim3((unsigned)((work + T - 1) / T)), dim3(T), 0, n->stream, in(0).p, n->gd.prelu[gl.slopes], o.p, o.c, o.cpad, npix);
            break;
        }
        case GLayer::PIXELSHUFFLE: {
            const GBuf& a = in(0);
            const size_t work = (size_t)o.h * o.w * o.c;
            hipLaunchKernelGGL(g_pixelshuffle, dim3((unsigned)((work + T - 1) / T)), dim3(T), 0, n->stream, a.p, a.h, a.w, a.cpad, o.p, o.c, o.cpad, gl.factor);
            break;
        }
        default: return fail("generic executor: unhandled layer kind");
        }
        HIP_TRY(hipGetLastError());
        finish_layer(ps);
        }
    }
    const int T2 = 256;
    for (PlaneState& ps : P) {
        const PlaneJob& j = ps.j;
        const GBuf& res = ps.buf[root(g.out_blob)];
        const int s = g.scale;
        if (f32) hipLaunchKernelGGL(g_output_f32, dim3((j.w * s + T2 - 1) / T2, j.h * s), dim3(T2), 0, n->stream, res.p, j.h * s, j.w * s, res.cpad, (float*)j.dst);
        else if (out_conv >= 0) { /* written by the last convolution */ }
        else if (j.cx1 > j.cx0 && j.cy1 > j.cy0)
            hipLaunchKernelGGL(g_output_u8, dim3(((j.cx1 - j.cx0) * s + T2 - 1) / T2, (j.cy1 - j.cy0) * s), dim3(T2), 0, n->stream, res.p, j.h * s, j.w * s, res.cpad,
                               (uint8_t*)j.dst, j.dst_stride, j.sy0 * s, j.sx0 * s, j.cy0 * s, j.cy1 * s, j.cx0 * s, j.cx1 * s);
        HIP_TRY(hipGetLastError());
        done_with(ps, g.out_blob);
    }
    return 0;
}

int generic_run_plane(uva_net* n, bool f32, const void* src, size_t src_stride, int sy0, int sx0, int h, int w, void* dst,
                      size_t dst_stride, int cy0, int cy1, int cx0, int cx1)
{
    return generic_run_planes(n, f32, std::vector<PlaneJob>{PlaneJob{src, src_stride, sy0, sx0, h, w, dst, dst_stride, cy0, cy1, cx0, cx1}});
}

// which planes of a frame go through the graph together: batch_of[i] = the batch of plane i, batches numbered in the order
// they run.  Planes of a batch are of one class (generic_plane_class), at most GEN_MAX_PLANES and `batch_pixels` input
// pixels (a single plane may exceed the bound: it is then a batch of its own).
void generic_plan_batches(const std::vector<PlaneDesc>& planes, bool batch_on, long long batch_pixels, std::vector<int>& batch_of)
{
    std::vector<int> cls, count;
    std::vector<long long> px;
    batch_of.clear();
    for (const PlaneDesc& p : planes) {
        const int c = generic_plane_class(p.w);
        const long long ppx = (long long)p.h * p.w;
        size_t k = 0;
        for (; batch_on && k < cls.size(); ++k)
            if (cls[k] == c && count[k] < GEN_MAX_PLANES && px[k] + ppx <= batch_pixels) break;
        if (!batch_on || k == cls.size()) { cls.push_back(c); count.push_back(0); px.push_back(0); k = cls.size() - 1; }
        ++count[k];
        px[k] += ppx;
        batch_of.push_back((int)k);
    }
}
// UVA_GENERIC_BATCH=0: one plane after the other (the A/B switch).  UVA_GENERIC_BATCH_PIXELS: a batch holds every array of
// its planes at once (at 4x, 16x the plane's pixels x 64 channels), so it is bounded to the planes of one 1080p frame
// (2.13 Mpixel) -- measured with 4x_Valar_v1 at 3840x2160 (12 planes), same bytes every way: this bound 0.343 s per frame
// and 33 GB in use, 4.2 Mpixel 0.345 s and 61 GB, all planes in one batch 0.338 s and 64 GB, one plane after the other
// 0.361 s and 33 GB
bool generic_batch_on() { static const bool on = [] { const char* e = uva::debug_env("UVA_GENERIC_BATCH"); return !e || std::atoi(e) != 0; }(); return on; }
long long generic_batch_pixels() { static const long long v = [] { const char* e = uva::debug_env("UVA_GENERIC_BATCH_PIXELS"); return e ? std::atoll(e) : 2200000ll; }(); return v; }

// the u8 frame call for a generic graph: every reference tile (upscale_processing.py:499-516) is one plane; planes that
// take the same kernels go through the graph together
int generic_process_u8_device(uva_net* n, const void* d_in, int h, int w, size_t in_stride, void* d_out, size_t out_stride,
                              int tile_size, int border)
{
    std::vector<PlaneDesc> planes;
    if (tile_size <= 0) { tile_size = 0; border = 0; }
    if (build_planes(h, w, tile_size, border, planes)) return 1;
    std::vector<int> batch_of;
    generic_plan_batches(planes, generic_batch_on(), generic_batch_pixels(), batch_of);
    std::vector<std::vector<PlaneJob>> batches;
    for (size_t i = 0; i < planes.size(); ++i) {
        const PlaneDesc& p = planes[i];
        if ((size_t)batch_of[i] >= batches.size()) batches.resize(batch_of[i] + 1);
        batches[batch_of[i]].push_back(PlaneJob{d_in, in_stride, p.src_y0, p.src_x0, p.h, p.w, d_out, out_stride, p.core_y0,
                                                std::min(p.core_y1, p.h), p.core_x0, std::min(p.core_x1, p.w)});
    }
    for (const auto& b : batches)
        if (generic_run_planes(n, false, b)) return 1;
    return 0;
}

int check_dims(const uva_net* n, int h, int w)
{
    if (!n) return fail("null net");
    if (h <= 0 || w <= 0 || (long long)h * w > (1ll << 28)) return fail("bad image size");
    return 0;
}

}  // namespace

// ---- `-m n=K`: non-local-means denoise (upscale/upscale_processing.py:350-361) -------------------------
namespace {

// ---- PNG encoding on the device (uva_png.hip.h) ---------------------------------------------------
struct PngDev {
    uint32_t* d_code = nullptr;
    uint8_t* d_hdr = nullptr;
    uint32_t* d_crcmul = nullptr;
    bool attr_set = false;
    // the synchronous utility's own buffers
    hipStream_t stream = nullptr;
    uint8_t *d_frame = nullptr, *d_blocks = nullptr;
    size_t d_frame_cap = 0, d_blocks_cap = 0;
};
std::mutex g_png_mu;          // the table uploads
std::mutex g_png_util_mu;     // uva_png_deflate_u8's stream and frame buffer (never taken inside g_png_mu, or the reverse)
PngDev g_png[16];

void png_release_all()
{
    std::lock_guard<std::mutex> lk2(g_png_util_mu);
    std::lock_guard<std::mutex> lk(g_png_mu);
    for (int d = 0; d < 16; ++d) {
        PngDev& c = g_png[d];
        if (!c.d_code && !c.stream) continue;
        (void)hipSetDevice(d);
        if (c.stream) { (void)hipStreamSynchronize(c.stream); (void)hipStreamDestroy(c.stream); }
        if (c.d_code) (void)hipFree(c.d_code);
        if (c.d_hdr) (void)hipFree(c.d_hdr);
        if (c.d_crcmul) (void)hipFree(c.d_crcmul);
        if (c.d_frame) (void)hipFree(c.d_frame);
        if (c.d_blocks) (void)hipFree(c.d_blocks);
        c = PngDev();
    }
}

// the code tables of this device (uploaded on first use); the caller holds no lock
int png_dev(int device, PngDev** out)
{
    if (device < 0 || device >= 16) return fail("bad device");
    std::lock_guard<std::mutex> lk(g_png_mu);
    PngDev& c = g_png[device];
    if (!c.d_code) {
        const PngTables& T = png_tables();
        HIP_TRY(hipSetDevice(device));
        HIP_TRY(hipMalloc((void**)&c.d_code, sizeof T.code));
        HIP_TRY(hipMalloc((void**)&c.d_hdr, sizeof T.hdr));
        HIP_TRY(hipMemcpy(c.d_code, T.code, sizeof T.code, hipMemcpyHostToDevice));
        HIP_TRY(hipMemcpy(c.d_hdr, T.hdr, sizeof T.hdr, hipMemcpyHostToDevice));
        HIP_TRY(hipMalloc((void**)&c.d_crcmul, sizeof T.crcmul));
        HIP_TRY(hipMemcpy(c.d_crcmul, T.crcmul, sizeof T.crcmul, hipMemcpyHostToDevice));
        HIP_TRY(hipFuncSetAttribute((const void*)png_deflate_kernel, hipFuncAttributeMaxDynamicSharedMemorySize, png_lds_bytes()));
        c.attr_set = true;
    }
    *out = &c;
    return 0;
}

// deflate the h x w u8 BGR frame at d_frame (HBM) into blocks at d_blocks (HBM, png_workspace_bytes(h, w)), on `stream`
int png_launch(int device, hipStream_t stream, const uint8_t* d_frame, size_t stride, int h, int w, uint8_t* d_blocks)
{
    if (h <= 0 || w <= 0 || 3 * (long long)w + 1 > PNG_FILT_CAP) return fail("PNG encoder: frame width out of range");
    PngDev* c = nullptr;
    if (png_dev(device, &c)) return 1;
    PngArgs a;
    std::memset(&a, 0, sizeof a);
    a.src = d_frame; a.stride = stride; a.h = h; a.w = w;
    a.rows_per_block = png_rows_per_block(w);
    a.nblocks = png_num_blocks(h, w);
    a.code = c->d_code; a.hdr = c->d_hdr; a.crcmul = c->d_crcmul;
    for (int t = 0; t < PNG_TABLES; ++t) a.hdr_bits[t] = png_tables().hdr_bits[t];
    a.meta = (uint32_t*)d_blocks;
    a.slots = d_blocks + png_meta_bytes(h, w);
    hipLaunchKernelGGL(png_deflate_kernel, dim3(a.nblocks), dim3(PNG_THREADS), png_lds_bytes(), stream, a);
    HIP_TRY(hipGetLastError());
    return 0;
}

// device layout of a frame's encoder output: [png_workspace_bytes(h, w): meta + slots][the same bound again: the packed bytes]
size_t png_device_bytes(int h, int w) { return 2 * png_workspace_bytes(h, w); }
uint8_t* png_packed(uint8_t* d_blocks, int h, int w) { return d_blocks + png_workspace_bytes(h, w); }

// the blocks at d_blocks, concatenated behind them (a device-to-device copy: microseconds), on `stream`
int png_pack_launch(hipStream_t stream, uint8_t* d_blocks, int h, int w)
{
    PngPackArgs a;
    a.meta = (const uint32_t*)d_blocks;
    a.slots = d_blocks + png_meta_bytes(h, w);
    a.nblocks = png_num_blocks(h, w);
    a.out_meta = nullptr;
    a.out_data = png_packed(d_blocks, h, w);
    hipLaunchKernelGGL(png_pack_kernel, dim3(a.nblocks), dim3(PNG_PACK_THREADS), 0, stream, a);
    HIP_TRY(hipGetLastError());
    return 0;
}

// meta + the first `bytes` packed bytes -> the page-locked workspace, by the copy engine, on `stream`
int png_download(hipStream_t stream, const uint8_t* d_blocks, int h, int w, void* ws, size_t bytes)
{
    const size_t mb = png_meta_bytes(h, w);
    HIP_TRY(hipMemcpyAsync(ws, d_blocks, mb, hipMemcpyDeviceToHost, stream));
    if (bytes) HIP_TRY(hipMemcpyAsync((uint8_t*)ws + mb, d_blocks + png_workspace_bytes(h, w), bytes, hipMemcpyDeviceToHost, stream));
    return 0;
}

// packed bytes of the frame whose meta is in the workspace
size_t png_total_bytes(const void* ws, int h, int w)
{
    const uint32_t* m = (const uint32_t*)ws;
    size_t t = 0;
    for (int b = 0, nb = png_num_blocks(h, w); b < nb; ++b) t += m[(size_t)b * PNG_META_WORDS];
    return t;
}

struct DenoiseCtx {
    hipStream_t stream = nullptr;
    uint8_t *d_in = nullptr, *d_out = nullptr, *d_l = nullptr, *d_l2 = nullptr, *d_ab = nullptr, *d_ab2 = nullptr;
    size_t cap_px = 0;
    int* d_table[2] = {nullptr, nullptr};
    int table_size[2] = {0, 0};
    float table_h[2] = {-1.f, -1.f};
    LabTables* d_lab = nullptr;          // OpenCV's 8-bit Lab tables (uva_denoise.hip.h)
};
std::mutex g_denoise_mu;
DenoiseCtx g_denoise[16];

// OpenCV fast_nlmeans_denoising_invoker.hpp: almost_dist2weight_ for 8-bit samples, squared distance
void nlm_weight_table(float h, int cn, std::vector<int>& table)
{
    const int search = 2 * NLM_S + 1, templ = 2 * NLM_T + 1;
    const int fixed_point_mult = INT_MAX / (search * search * 255);
    const double mult = (double)(1 << NLM_SHIFT) / (templ * templ);
    const int max_dist = 255 * 255 * cn;
    const int almost_max = (int)(max_dist / mult + 1);
    table.resize(almost_max);
    for (int ad = 0; ad < almost_max; ++ad) {
        const double dist = ad * mult;
        double w = std::exp(-dist / ((double)h * h * cn));
        if (std::isnan(w)) w = 1.0;
        int weight = (int)std::lrint(fixed_point_mult * w);
        if (weight < 0.001 * fixed_point_mult) weight = 0;
        table[ad] = weight;
    }
}

void denoise_release_all()
{
    std::lock_guard<std::mutex> lk(g_denoise_mu);
    for (int d = 0; d < 16; ++d) {
        DenoiseCtx& c = g_denoise[d];
        if (!c.stream && !c.d_in && !c.d_table[0] && !c.d_table[1]) continue;
        (void)hipSetDevice(d);
        if (c.stream) (void)hipStreamSynchronize(c.stream);
        for (uint8_t* p : {c.d_in, c.d_out, c.d_l, c.d_l2, c.d_ab, c.d_ab2})
            if (p) (void)hipFree(p);
        for (int* t : c.d_table)
            if (t) (void)hipFree(t);
        if (c.d_lab) (void)hipFree(c.d_lab);
        if (c.stream) (void)hipStreamDestroy(c.stream);
        c = DenoiseCtx();
    }
}

int denoise_ctx(int device, size_t px, DenoiseCtx** out)
{
    int count = 0;
    if (hipGetDeviceCount(&count) != hipSuccess || count <= 0) return fail("no HIP device available: libuva has no CPU path");
    if (device < 0 || device >= count || device >= 16) return fail("HIP device " + std::to_string(device) + " does not exist");
    HIP_TRY(hipSetDevice(device));
    DenoiseCtx& c = g_denoise[device];
    if (!c.stream) HIP_TRY(hipStreamCreateWithFlags(&c.stream, hipStreamNonBlocking));
    if (!c.d_lab) {
        static const LabTables host_tables = [] { LabTables t; lab_tables_host(t); return t; }();
        HIP_TRY(hipMalloc((void**)&c.d_lab, sizeof(LabTables)));
        HIP_TRY(hipMemcpy(c.d_lab, &host_tables, sizeof(LabTables), hipMemcpyHostToDevice));
    }
    if (c.cap_px < px) {
        // uva_denoise_u8_device is asynchronous: frames queued on the stream may still use the buffers about to go
        // (hipFree happens to synchronise the device; said here instead of relied on)
        if (c.cap_px) HIP_TRY(hipStreamSynchronize(c.stream));
        for (uint8_t** p : {&c.d_in, &c.d_out, &c.d_l, &c.d_l2, &c.d_ab, &c.d_ab2}) {
            if (*p) (void)hipFree(*p);
            *p = nullptr;
        }
        c.cap_px = 0;
        HIP_TRY(hipMalloc((void**)&c.d_in, px * 3));
        HIP_TRY(hipMalloc((void**)&c.d_out, px * 3));
        HIP_TRY(hipMalloc((void**)&c.d_l, px));
        HIP_TRY(hipMalloc((void**)&c.d_l2, px));
        HIP_TRY(hipMalloc((void**)&c.d_ab, px * 2));
        HIP_TRY(hipMalloc((void**)&c.d_ab2, px * 2));
        c.cap_px = px;
    }
    *out = &c;
    return 0;
}

int denoise_table(DenoiseCtx& c, int which, float h, int cn)
{
    if (c.table_h[which] == h && c.d_table[which]) return 0;
    std::vector<int> t;
    nlm_weight_table(h, cn, t);
    if (c.d_table[which]) (void)hipFree(c.d_table[which]);
    c.d_table[which] = nullptr;
    HIP_TRY(hipMalloc((void**)&c.d_table[which], t.size() * sizeof(int)));
    HIP_TRY(hipMemcpy(c.d_table[which], t.data(), t.size() * sizeof(int), hipMemcpyHostToDevice));
    c.table_size[which] = (int)t.size();
    c.table_h[which] = h;
    return 0;
}

}  // namespace

extern "C" {

static int denoise_launch(DenoiseCtx* c, const uint8_t* d_src, size_t src_stride, uint8_t* d_dst, size_t dst_stride, int h, int w);

int uva_denoise_u8(int device, const uint8_t* in, int h, int w, size_t in_stride, uint8_t* out, size_t out_stride,
                   float h_luma, float h_color)
{
    if (!in || !out || h <= 0 || w <= 0 || (long long)h * w > (1ll << 28)) return fail("bad image");
    if (in_stride < (size_t)w * 3 || out_stride < (size_t)w * 3) return fail("row stride too small");
    if (!(h_luma > 0.f) || !(h_color > 0.f)) return fail("denoise strength must be positive");
    std::lock_guard<std::mutex> lk(g_denoise_mu);
    DenoiseCtx* c = nullptr;
    const size_t px = (size_t)h * w;
    if (denoise_ctx(device, px, &c)) return 1;
    if (denoise_table(*c, 0, h_luma, 1) || denoise_table(*c, 1, h_color, 2)) return 1;
    HIP_TRY(hipMemcpy2DAsync(c->d_in, (size_t)w * 3, in, in_stride, (size_t)w * 3, h, hipMemcpyHostToDevice, c->stream));
    if (denoise_launch(c, c->d_in, (size_t)w * 3, c->d_out, (size_t)w * 3, h, w)) return 1;
    HIP_TRY(hipMemcpy2DAsync(out, out_stride, c->d_out, (size_t)w * 3, (size_t)w * 3, h, hipMemcpyDeviceToHost, c->stream));
    HIP_TRY(hipStreamSynchronize(c->stream));
    return 0;
}

// The four kernels of the stage on the context's stream, frame in d_src (strided), result in d_dst (strided)
static int denoise_launch(DenoiseCtx* c, const uint8_t* d_src, size_t src_stride, uint8_t* d_dst, size_t dst_stride, int h, int w)
{
    const dim3 rows((w + 255) / 256, h), tiles((w + NLM_BLK - 1) / NLM_BLK, (h + NLM_BLK - 1) / NLM_BLK);
    hipLaunchKernelGGL(nlm_bgr2lab, rows, dim3(256), 0, c->stream, d_src, src_stride, h, w, c->d_l, c->d_ab, c->d_lab);
    hipLaunchKernelGGL(nlm_plane<1>, tiles, dim3(NLM_BLK * NLM_BLK), 0, c->stream, c->d_l, h, w, c->d_table[0], c->table_size[0], c->d_l2);
    hipLaunchKernelGGL(nlm_plane<2>, tiles, dim3(NLM_BLK * NLM_BLK), 0, c->stream, c->d_ab, h, w, c->d_table[1], c->table_size[1], c->d_ab2);
    hipLaunchKernelGGL(nlm_lab2bgr, rows, dim3(256), 0, c->stream, c->d_l2, c->d_ab2, h, w, d_dst, dst_stride, c->d_lab);
    HIP_TRY(hipGetLastError());
    return 0;
}

int uva_denoise_u8_device(int device, const void* d_in, int h, int w, size_t in_stride, void* d_out, size_t out_stride,
                          float h_luma, float h_color, uva_net* after, uva_net* before)
{
    if (!d_in || !d_out || h <= 0 || w <= 0 || (long long)h * w > (1ll << 28)) return fail("bad image");
    if (in_stride < (size_t)w * 3 || out_stride < (size_t)w * 3) return fail("row stride too small");
    if (!(h_luma > 0.f) || !(h_color > 0.f)) return fail("denoise strength must be positive");
    for (uva_net* n : {after, before})
        if (n) {
            if (ensure_device(n)) return 1;
            if (n->device != device) return fail("uva_denoise_u8_device: the net is on another device");
        }
    std::lock_guard<std::mutex> lk(g_denoise_mu);
    DenoiseCtx* c = nullptr;
    if (denoise_ctx(device, (size_t)h * w, &c)) return 1;
    // the weight tables are uploaded with a blocking copy: the stream must not be reading the old ones
    if (c->table_h[0] != h_luma || c->table_h[1] != h_color) HIP_TRY(hipStreamSynchronize(c->stream));
    if (denoise_table(*c, 0, h_luma, 1) || denoise_table(*c, 1, h_color, 2)) return 1;
    if (after) {                 // what `after` has been asked to do so far (it wrote d_in, or still reads d_out) comes first
        SyncEventScope ev(after);
        if (!ev.e) return 1;
        HIP_TRY(hipEventRecord(ev.e, after->stream));
        HIP_TRY(hipStreamWaitEvent(c->stream, ev.e, 0));
    }
    if (denoise_launch(c, (const uint8_t*)d_in, in_stride, (uint8_t*)d_out, out_stride, h, w)) return 1;
    if (before) {                // ... and whatever `before` is asked to do from now on comes after this frame
        SyncEventScope ev(before);
        if (!ev.e) return 1;
        HIP_TRY(hipEventRecord(ev.e, c->stream));
        HIP_TRY(hipStreamWaitEvent(before->stream, ev.e, 0));
    }
    return 0;
}

int uva_denoise_synchronize(int device)
{
    std::lock_guard<std::mutex> lk(g_denoise_mu);
    if (device < 0 || device >= 16) return fail("bad device");
    if (!g_denoise[device].stream) return 0;
    HIP_TRY(hipSetDevice(device));
    HIP_TRY(hipStreamSynchronize(g_denoise[device].stream));
    return 0;
}

int uva_debug_denoise_stage(int device, int stage, const uint8_t* in, int h, int w, float strength, uint8_t* out)
{
    if (!in || !out || h <= 0 || w <= 0) return fail("bad argument");
    std::lock_guard<std::mutex> lk(g_denoise_mu);
    DenoiseCtx* c = nullptr;
    const size_t px = (size_t)h * w;
    if (denoise_ctx(device, px, &c)) return 1;
    const dim3 rows((w + 255) / 256, h), tiles((w + NLM_BLK - 1) / NLM_BLK, (h + NLM_BLK - 1) / NLM_BLK);
    if (stage == 0) {          // bgr [h][w][3] -> Lab interleaved [h][w][3]
        HIP_TRY(hipMemcpy(c->d_in, in, px * 3, hipMemcpyHostToDevice));
        hipLaunchKernelGGL(nlm_bgr2lab, rows, dim3(256), 0, c->stream, c->d_in, (size_t)w * 3, h, w, c->d_l, c->d_ab, c->d_lab);
        HIP_TRY(hipStreamSynchronize(c->stream));
        std::vector<uint8_t> l(px), ab(px * 2);
        HIP_TRY(hipMemcpy(l.data(), c->d_l, px, hipMemcpyDeviceToHost));
        HIP_TRY(hipMemcpy(ab.data(), c->d_ab, px * 2, hipMemcpyDeviceToHost));
        for (size_t i = 0; i < px; ++i) { out[3 * i] = l[i]; out[3 * i + 1] = ab[2 * i]; out[3 * i + 2] = ab[2 * i + 1]; }
    } else if (stage == 1) {   // Lab interleaved -> bgr
        std::vector<uint8_t> l(px), ab(px * 2);
        for (size_t i = 0; i < px; ++i) { l[i] = in[3 * i]; ab[2 * i] = in[3 * i + 1]; ab[2 * i + 1] = in[3 * i + 2]; }
        HIP_TRY(hipMemcpy(c->d_l, l.data(), px, hipMemcpyHostToDevice));
        HIP_TRY(hipMemcpy(c->d_ab, ab.data(), px * 2, hipMemcpyHostToDevice));
        hipLaunchKernelGGL(nlm_lab2bgr, rows, dim3(256), 0, c->stream, c->d_l, c->d_ab, h, w, c->d_out, (size_t)w * 3, c->d_lab);
        HIP_TRY(hipStreamSynchronize(c->stream));
        HIP_TRY(hipMemcpy(out, c->d_out, px * 3, hipMemcpyDeviceToHost));
    } else if (stage == 2 || stage == 3) {   // NLM on a 1-channel (2) / 2-channel (3) u8 image
        const int cn = stage - 1;
        if (!(strength > 0.f)) return fail("denoise strength must be positive");
        if (denoise_table(*c, cn - 1, strength, cn)) return 1;
        uint8_t* src = cn == 1 ? c->d_l : c->d_ab;
        uint8_t* dst = cn == 1 ? c->d_l2 : c->d_ab2;
        HIP_TRY(hipMemcpy(src, in, px * cn, hipMemcpyHostToDevice));
        if (cn == 1) hipLaunchKernelGGL(nlm_plane<1>, tiles, dim3(NLM_BLK * NLM_BLK), 0, c->stream, src, h, w, c->d_table[0], c->table_size[0], dst);
        else hipLaunchKernelGGL(nlm_plane<2>, tiles, dim3(NLM_BLK * NLM_BLK), 0, c->stream, src, h, w, c->d_table[1], c->table_size[1], dst);
        HIP_TRY(hipGetLastError());
        HIP_TRY(hipStreamSynchronize(c->stream));
        HIP_TRY(hipMemcpy(out, dst, px * cn, hipMemcpyDeviceToHost));
    } else {
        return fail("bad stage");
    }
    HIP_TRY(hipGetLastError());
    return 0;
}

}  // extern "C"

extern "C" {

int uva_abi_version(void) { return UVA_ABI_VERSION; }
const char* uva_last_error(void) { return g_err.c_str(); }

int uva_get_gpu_count(void)
{
    int count = 0;
    if (hipGetDeviceCount(&count) != hipSuccess) return 0;
    return count;
}

int uva_get_default_gpu_index(void) { return uva_get_gpu_count() > 0 ? 0 : -1; }

int uva_get_gpu_info(int index, int* type, char* name, size_t name_len)
{
    const int count = uva_get_gpu_count();
    if (index < 0 || index >= count) return fail("no such HIP device");
    hipDeviceProp_t prop;
    HIP_TRY(hipGetDeviceProperties(&prop, index));
    if (type) *type = prop.integrated ? 1 : 0;
    if (name && name_len) std::snprintf(name, name_len, "%s (%s)", prop.name, prop.gcnArchName);
    return 0;
}

int uva_get_gpu_pci_bus_id(int index, char* out, size_t out_len)
{
    if (!out || out_len < 13) return fail("buffer too small for a PCI address");
    const int count = uva_get_gpu_count();
    if (index < 0 || index >= count) return fail("no such HIP device");
    HIP_TRY(hipDeviceGetPCIBusId(out, (int)out_len, index));
    return 0;
}

void uva_destroy_gpu_instance(void)
{
    {
        std::lock_guard<std::mutex> lk(g_nets_mu);
        for (uva_net* n : g_nets) n->free_device();
    }
    denoise_release_all();
    png_release_all();
}

uva_net* uva_net_create(void)
{
    uva_net* n = new uva_net();
    std::lock_guard<std::mutex> lk(g_nets_mu);
    g_nets.insert(n);
    return n;
}

void uva_net_destroy(uva_net* n)
{
    if (!n) return;
    {
        std::lock_guard<std::mutex> lk(g_nets_mu);
        g_nets.erase(n);
    }
    n->free_device();
    delete n;
}

int uva_net_set_device(uva_net* n, int device)
{
    if (!n) return fail("null net");
    if (device < 0) return fail("negative device index: libuva has no CPU path");
    if (n->dev_ready && device != n->device) n->free_device();
    n->device = device;
    return 0;
}

int uva_net_load_param(uva_net* n, const char* path)
{
    if (!n || !path) return fail("null argument");
    n->free_device();
    n->generic = false;
    n->gg = GenericGraph();
    std::string err;
    // UVA_GENERIC=1 (tests): run even the SRVGGNetCompact graphs through the generic executor
    const char* force = uva::debug_env("UVA_GENERIC");
    if (!(force && std::atoi(force)) && parse_param(path, n->g, err)) return 0;
    // not the SRVGGNetCompact pattern the fused kernels are written for: the generic executor, if every layer
    // type is one it knows (4x_Valar_v1 is); otherwise the first parser's message stands
    std::string gerr;
    if (!parse_param_generic(path, n->gg, gerr)) return fail(gerr.find("unsupported layer type") != std::string::npos ? gerr : err);
    n->generic = true;
    return 0;
}

int uva_net_load_model(uva_net* n, const char* path)
{
    if (!n || !path) return fail("null argument");
    n->free_device();
    std::string err;
    if (n->generic ? !load_bin_generic(path, n->gg, err) : !load_bin(path, n->g, err)) return fail(err);
    return 0;
}

int uva_net_device(const uva_net* n) { return n ? n->device : -1; }

int uva_net_scale(const uva_net* n)
{
    if (!n) return 0;
    if (n->generic) return n->gg.param_loaded ? n->gg.scale : 0;
    return n->g.param_loaded ? n->g.scale : 0;
}
int uva_net_num_features(const uva_net* n)
{
    if (!n) return 0;
    if (n->generic) return n->gg.param_loaded ? n->gg.max_channels : 0;
    return n->g.param_loaded ? n->g.nf : 0;
}
int uva_net_num_convs(const uva_net* n)
{
    if (!n) return 0;
    if (n->generic) return n->gg.param_loaded ? (int)n->gg.convs.size() : 0;
    return n->g.param_loaded ? (int)n->g.convs.size() : 0;
}

int uva_net_debug_generic_plan(const uva_net* n, int* info)
{
    if (!n || !info) return fail("null argument");
    if (!n->generic || !n->gg.param_loaded) return fail("not a generic graph");
    const GenericGraph& g = n->gg;
    info[0] = (int)g.group_channels.size();
    info[1] = info[2] = info[3] = 0;
    for (const GLayer& l : g.layers)
        if (l.kind == GLayer::CONCAT) { info[1]++; info[2] += l.concat_mode == 2; info[3] += l.concat_mode == 1; }
    int lds = 0;
    for (const GLayer& l : g.layers)
        if (l.kind == GLayer::CONV && l.ksize == 3) {
            const ConvWeights& c = g.convs[l.conv];
            const int cin_pad = (c.cin + 31) / 32 * 32, cout_pad = (c.cout + 15) / 16 * 16;
            lds += cout_pad <= 64 && cin_pad <= 192 && g_conv3_lds_bytes(cin_pad, cout_pad / 16) <= 160 * 1024;
        }
    info[4] = lds;
    info[5] = 0;
    for (int c : g.group_channels) info[5] = std::max(info[5], c);
    info[6] = (int)find_rdbs(g).size();
    return 0;
}

int uva_net_synchronize(uva_net* n)
{
    if (!n) return fail("null net");
    if (!n->dev_ready) return 0;
    HIP_TRY(hipSetDevice(n->device));
    HIP_TRY(hipStreamSynchronize(n->stream));
    if (n->s_d2h) HIP_TRY(hipStreamSynchronize(n->s_d2h));   // in-flight pipelined frames (results stay collectable)
    resolve_events(n);
    return 0;
}

int uva_net_wait_for(uva_net* n, uva_net* producer)
{
    if (!n || !producer) return fail("null net");
    if (n == producer || !producer->dev_ready) return 0;
    if (ensure_device(n)) return 1;
    if (producer->device != n->device) return fail("uva_net_wait_for: nets are on different devices");
    SyncEventScope ev(n);           // (recycled at the end of the scope: the wait has captured the recorded state)
    if (!ev.e) return 1;
    HIP_TRY(hipEventRecord(ev.e, producer->stream));
    HIP_TRY(hipStreamWaitEvent(n->stream, ev.e, 0));
    return 0;
}

int uva_net_process_u8_device(uva_net* n, const void* d_in, int h, int w, size_t in_stride, void* d_out,
                              size_t out_stride, int tile_size, int border)
{
    if (check_dims(n, h, w)) return 1;
    if (!d_in || !d_out) return fail("null frame pointer");
    if (ensure_device(n)) return 1;
    if (in_stride < (size_t)w * 3 || out_stride < (size_t)w * uva_net_scale(n) * 3) return fail("row stride too small");
    // the tail kernels address the residual bytes with 32-bit offsets from the frame's base (the LDS-DMA's lane offset is 32-bit)
    if ((unsigned long long)in_stride * (unsigned long long)h >= (1ull << 32) - 64 ||
        (unsigned long long)out_stride * (unsigned long long)h * (unsigned long long)uva_net_scale(n) >= (1ull << 32) - 64)
        return fail("frame of 4 GB or more");
    if (n->generic) return generic_process_u8_device(n, d_in, h, w, in_stride, d_out, out_stride, tile_size, border);
    Workspace* ws = nullptr;
    if (get_workspace(n, h, w, tile_size, border, &ws)) return 1;
    n->last.valid = true; n->last.ws = ws; n->last.f32 = false; n->last.src = d_in; n->last.src_stride = in_stride;
    n->last.dst = d_out; n->last.dst_stride = out_stride;
    return run_graph(n, ws, false, d_in, in_stride, d_out, out_stride, -1);
}

// `count` frames of ONE geometry, all resident on the net's device.  The 1x net takes up to S10_MAXB of them per launch
// (sub10_kernel: the segments' warm-up rows and the pipeline's fill and drain are paid once per launch, not once per frame);
// every other net -- and whatever does not fit the kernel's row table -- runs frame by frame, exactly as `count` calls of
// uva_net_process_u8_device would.  The bytes are those of the single-frame calls either way (tests).
int uva_net_process_u8_device_batch(uva_net* n, const void* const* d_in, void* const* d_out, int count, int h, int w,
                                    size_t in_stride, size_t out_stride, int tile_size, int border)
{
    if (check_dims(n, h, w)) return 1;
    if (count < 0 || (count > 0 && (!d_in || !d_out))) return fail("bad frame list");
    for (int i = 0; i < count; ++i)
        if (!d_in[i] || !d_out[i]) return fail("null frame pointer");
    if (count == 0) return 0;
    if (ensure_device(n)) return 1;
    if (in_stride < (size_t)w * 3 || out_stride < (size_t)w * uva_net_scale(n) * 3) return fail("row stride too small");
    Workspace* ws = nullptr;
    if (!n->generic && get_workspace(n, h, w, tile_size, border, &ws)) return 1;
    int i = 0;
    if (ws && sub10_route(n, ws, false, -1) && !n->split5) {
        if ((unsigned long long)in_stride * (unsigned long long)h >= (1ull << 32) - 64 ||
            (unsigned long long)out_stride * (unsigned long long)h >= (1ull << 32) - 64)
            return fail("frame of 4 GB or more");
        while (i < count) {
            const int k = std::min(count - i, std::max(1, ws->sub10_max_batch));
            const int rc = run_sub10(n, ws, d_in + i, in_stride, d_out + i, out_stride, k);
            if (rc == 1) return 1;
            if (rc == 2) {
                if (k == 1) break;              // not even one frame fits: the rest goes frame by frame below
                continue;                       // (launch_sub10 lowered sub10_max_batch)
            }
            i += k;
            n->last.valid = true; n->last.ws = ws; n->last.f32 = false; n->last.src = d_in[i - 1]; n->last.src_stride = in_stride;
            n->last.dst = d_out[i - 1]; n->last.dst_stride = out_stride;
        }
    }
    for (; i < count; ++i)
        if (uva_net_process_u8_device(n, d_in[i], h, w, in_stride, d_out[i], out_stride, tile_size, border)) return 1;
    return 0;
}

// ---- pipelined host route -----------------------------------------------------------------------
namespace {
bool is_pinned_host(const void* p)
{
    hipPointerAttribute_t a;
    if (hipPointerGetAttributes(&a, p) != hipSuccess) {
        (void)hipGetLastError();   // an ordinary malloc'ed pointer: not an error
        return false;
    }
    return a.type == hipMemoryTypeHost;
}
}  // namespace

void* uva_host_alloc(size_t bytes)
{
    void* p = nullptr;
    if (hipHostMalloc(&p, bytes ? bytes : 1, hipHostMallocDefault) != hipSuccess) {
        fail("uva_host_alloc: hipHostMalloc failed");
        return nullptr;
    }
    return p;
}

void uva_host_free(void* p)
{
    if (p) (void)hipHostFree(p);
}

}  // extern "C"
namespace {
// uva_net_submit_u8 (png_ws == nullptr: the result frame goes to `out`) and uva_net_submit_u8_png (the result frame stays
// in HBM, its PNG deflate blocks go to the page-locked workspace png_ws)
long long submit_u8(uva_net* n, const uint8_t* in, int h, int w, size_t in_stride, uint8_t* out, size_t out_stride,
                    int tile_size, int border, void* png_ws, size_t png_ws_bytes)
{
    if (check_dims(n, h, w)) return -1;
    if (!in || (!out && !png_ws)) { fail("null frame pointer"); return -1; }
    if (png_ws && !is_pinned_host(png_ws)) { fail("PNG workspace must be page-locked host memory (uva_host_alloc)"); return -1; }
    if (ensure_device(n)) return -1;
    auto tryhip = [](hipError_t e, const char* what) { return e == hipSuccess ? 0 : fail(std::string(what) + ": " + hipGetErrorString(e)); };
    const int s = uva_net_scale(n);
    const size_t in_row = (size_t)w * 3, out_row = (size_t)w * s * 3;
    if (png_ws) out_stride = out_row;
    if (in_stride < in_row || out_stride < out_row) { fail("row stride too small"); return -1; }
    const size_t in_bytes = in_row * h, out_bytes = out_row * (size_t)h * s;
    uva_net::PipeSlot* free_slot = nullptr;
    for (auto& c : n->pipe)
        if (!c.busy) { free_slot = &c; break; }
    if (!free_slot) { fail("uva_net_submit_u8: " + std::to_string(uva_net::PIPE_SLOTS) + " frames already in flight, collect one first"); return -1; }
    uva_net::PipeSlot& ps = *free_slot;
    if (!n->s_h2d) {
        if (tryhip(hipStreamCreateWithFlags(&n->s_h2d, hipStreamNonBlocking), "hipStreamCreate") ||
            tryhip(hipStreamCreateWithFlags(&n->s_d2h, hipStreamNonBlocking), "hipStreamCreate")) return -1;
    }
    if (!ps.ev_h2d) {
        if (tryhip(hipEventCreateWithFlags(&ps.ev_h2d, hipEventDisableTiming), "hipEventCreate") ||
            tryhip(hipEventCreateWithFlags(&ps.ev_done, hipEventDisableTiming), "hipEventCreate") ||
            tryhip(hipEventCreateWithFlags(&ps.ev_d2h, hipEventDisableTiming), "hipEventCreate")) return -1;
    }
    if (grow_dev(&ps.d_in, &ps.d_in_cap, in_bytes) || grow_dev(&ps.d_out, &ps.d_out_cap, out_bytes)) return -1;
    if (png_ws) {
        if (uva_png_workspace_bytes(h * s, w * s) == 0) { fail("PNG encoder: frame width out of range"); return -1; }
        if (png_ws_bytes < png_workspace_bytes(h * s, w * s)) { fail("PNG workspace too small"); return -1; }
        if (grow_dev(&ps.d_png, &ps.d_png_cap, png_device_bytes(h * s, w * s))) return -1;
    }
    // H2D: straight from the caller's buffer when it is pinned (uva_host_alloc / hipHostMalloc /
    // hipHostRegister), through this slot's pinned staging buffer otherwise
    const uint8_t* src = in;
    size_t src_stride = in_stride;
    if (!is_pinned_host(in)) {
        if (grow_host(&ps.h_in, &ps.h_in_cap, in_bytes)) return -1;
        for (int y = 0; y < h; ++y) std::memcpy(ps.h_in + y * in_row, in + y * in_stride, in_row);
        src = ps.h_in; src_stride = in_row;
    }
    if (tryhip(hipMemcpy2DAsync(ps.d_in, in_row, src, src_stride, in_row, h, hipMemcpyHostToDevice, n->s_h2d), "H2D") ||
        tryhip(hipEventRecord(ps.ev_h2d, n->s_h2d), "hipEventRecord") ||
        tryhip(hipStreamWaitEvent(n->stream, ps.ev_h2d, 0), "hipStreamWaitEvent")) return -1;
    if (uva_net_process_u8_device(n, ps.d_in, h, w, in_row, ps.d_out, out_row, tile_size, border)) return -1;
    // png: the deflate kernel follows the net on its stream and leaves the blocks in HBM, packed end to end by a second
    // (device-to-device) kernel: 0.15 ms per 4K frame together
    if (png_ws && (png_launch(n->device, n->stream, ps.d_out, out_row, h * s, w * s, ps.d_png) ||
                   png_pack_launch(n->stream, ps.d_png, h * s, w * s))) return -1;
    if (tryhip(hipEventRecord(ps.ev_done, n->stream), "hipEventRecord") ||
        tryhip(hipStreamWaitEvent(n->s_d2h, ps.ev_done, 0), "hipStreamWaitEvent")) return -1;
    if (png_ws) {
        // ... and the copy engine takes them to the caller's page-locked workspace while the next frame computes.  How
        // many bytes there are is only known on the device: as many as the previous frame had (+ 6 %) go now, collect
        // fetches the rest should this frame be larger.  (A kernel writing to host memory instead keeps CUs busy for
        // the PCIe transfer: 0.24 ms per 4K frame that the next frame's net then waits for.)
        const size_t cap = png_workspace_bytes(h * s, w * s) - png_meta_bytes(h * s, w * s);
        const size_t guess = n->png_guess ? std::min(cap, n->png_guess + n->png_guess / 16 + 65536) : std::min(cap, out_bytes * 5 / 8);
        if (png_download(n->s_d2h, ps.d_png, h * s, w * s, png_ws, guess)) return -1;
        if (tryhip(hipEventRecord(ps.ev_d2h, n->s_d2h), "hipEventRecord")) return -1;
        ps.png_ws = (uint8_t*)png_ws; ps.png_sent = guess; ps.png_h = h * s; ps.png_w = w * s;
        ps.user_out = nullptr;
        ps.busy = true;
        ps.ticket = n->next_ticket;
        return n->next_ticket++;
    }
    uint8_t* dst = out;
    size_t dst_stride = out_stride;
    ps.user_out = nullptr;
    ps.png_ws = nullptr;
    if (!is_pinned_host(out)) {
        if (grow_host(&ps.h_out, &ps.h_out_cap, out_bytes)) return -1;
        dst = ps.h_out; dst_stride = out_row;
        ps.user_out = out; ps.user_out_stride = out_stride; ps.out_row = out_row; ps.out_rows = h * s;
    }
    if (ps.user_out) {
        const int rows = h * s, per = (rows + uva_net::PipeSlot::BANDS - 1) / uva_net::PipeSlot::BANDS;
        for (int k = 0; k < uva_net::PipeSlot::BANDS; ++k) {
            if (!ps.ev_band[k] && tryhip(hipEventCreateWithFlags(&ps.ev_band[k], hipEventDisableTiming), "hipEventCreate")) return -1;
            const int r0 = std::min(rows, k * per), nr = std::min(rows, r0 + per) - r0;
            if (nr > 0 && tryhip(hipMemcpyAsync(dst + (size_t)r0 * out_row, ps.d_out + (size_t)r0 * out_row, (size_t)nr * out_row,
                                                hipMemcpyDeviceToHost, n->s_d2h), "D2H")) return -1;
            if (tryhip(hipEventRecord(ps.ev_band[k], n->s_d2h), "hipEventRecord")) return -1;
        }
    } else if (tryhip(hipMemcpy2DAsync(dst, dst_stride, ps.d_out, out_row, out_row, (size_t)h * s, hipMemcpyDeviceToHost, n->s_d2h), "D2H")) {
        return -1;
    }
    if (tryhip(hipEventRecord(ps.ev_d2h, n->s_d2h), "hipEventRecord")) return -1;
    ps.busy = true;
    ps.ticket = n->next_ticket;
    return n->next_ticket++;
}
}  // namespace
extern "C" {

long long uva_net_submit_u8(uva_net* n, const uint8_t* in, int h, int w, size_t in_stride, uint8_t* out,
                            size_t out_stride, int tile_size, int border)
{
    return submit_u8(n, in, h, w, in_stride, out, out_stride, tile_size, border, nullptr, 0);
}

long long uva_net_submit_u8_png(uva_net* n, const uint8_t* in, int h, int w, size_t in_stride, void* png_ws, size_t png_ws_bytes,
                                int tile_size, int border)
{
    if (!png_ws) { fail("null PNG workspace"); return -1; }
    return submit_u8(n, in, h, w, in_stride, nullptr, 0, tile_size, border, png_ws, png_ws_bytes);
}

size_t uva_png_workspace_bytes(int h, int w)
{
    if (h <= 0 || w <= 0 || 3 * (long long)w + 1 > PNG_FILT_CAP) return 0;
    return png_workspace_bytes(h, w);
}

int uva_png_assemble(const void* png_ws, int h, int w, uint8_t* out, size_t cap, size_t* len)
{
    if (!png_ws || uva_png_workspace_bytes(h, w) == 0) return fail("bad argument");
    std::string err;
    if (png_assemble((const uint8_t*)png_ws, h, w, out, cap, len, err)) return fail(err);
    return 0;
}

int uva_png_deflate_u8(int device, const uint8_t* bgr, int h, int w, size_t stride, void* png_ws, size_t png_ws_bytes)
{
    if (!bgr || stride < (size_t)w * 3) return fail("bad frame");
    if (uva_png_workspace_bytes(h, w) == 0) return fail("PNG encoder: frame width out of range");
    if (!is_pinned_host(png_ws)) return fail("PNG workspace must be page-locked host memory (uva_host_alloc)");
    PngDev* c = nullptr;
    if (png_dev(device, &c)) return 1;
    std::lock_guard<std::mutex> lk(g_png_util_mu);
    HIP_TRY(hipSetDevice(device));
    if (!c->stream) HIP_TRY(hipStreamCreateWithFlags(&c->stream, hipStreamNonBlocking));
    const size_t row = (size_t)w * 3;
    if (png_ws_bytes < png_workspace_bytes(h, w)) return fail("PNG workspace too small");
    if (grow_dev(&c->d_frame, &c->d_frame_cap, row * h) || grow_dev(&c->d_blocks, &c->d_blocks_cap, png_device_bytes(h, w))) return 1;
    HIP_TRY(hipMemcpy2DAsync(c->d_frame, row, bgr, stride, row, h, hipMemcpyHostToDevice, c->stream));
    if (png_launch(device, c->stream, c->d_frame, row, h, w, c->d_blocks) || png_pack_launch(c->stream, c->d_blocks, h, w)) return 1;
    if (png_download(c->stream, c->d_blocks, h, w, png_ws, 0)) return 1;
    HIP_TRY(hipStreamSynchronize(c->stream));
    const size_t total = png_total_bytes(png_ws, h, w);
    if (total > png_workspace_bytes(h, w) - png_meta_bytes(h, w)) return fail("PNG encoder: block sizes out of range");
    HIP_TRY(hipMemcpy((uint8_t*)png_ws + png_meta_bytes(h, w), png_packed(c->d_blocks, h, w), total, hipMemcpyDeviceToHost));
    return 0;
}

int uva_png_decode_bgr(const uint8_t* file, size_t len, uint8_t* out, size_t cap, int* h, int* w)
{
    if (!file) return fail("null file image");
    std::string err;
    int rc;
    try {
        rc = png_read_bgr(file, len, out, cap, h, w, err);
    } catch (const std::exception& e) {              // out of memory: an error, not a crash across the C ABI
        return fail(std::string("PNG reader: ") + e.what());
    }
    if (rc == 1) return fail(err);
    if (rc == 2) { fail("PNG of a kind the fast reader does not take (16-bit, palette or interlaced)"); return 2; }
    return 0;
}

int uva_debug_zlib_decompress(const uint8_t* in, size_t n, uint8_t* out, size_t out_len)
{
    if (!in || (!out && out_len)) return fail("null argument");
    std::string err;
    if (zlib_decompress_exact(in, n, out, out_len, err)) return fail(err);
    return 0;
}

int uva_debug_png_deflate_host(const uint8_t* bgr, int h, int w, size_t stride, void* png_ws, size_t png_ws_bytes)
{
    if (!bgr || !png_ws || stride < (size_t)w * 3 || uva_png_workspace_bytes(h, w) == 0 || png_ws_bytes < png_workspace_bytes(h, w))
        return fail("bad argument");
    png_deflate_host(bgr, stride, h, w, (uint8_t*)png_ws);
    return 0;
}

int uva_net_collect_u8(uva_net* n, long long ticket)
{
    if (!n || ticket < 0) return fail("bad ticket");
    uva_net::PipeSlot* slot = nullptr;
    for (auto& c : n->pipe)
        if (c.busy && c.ticket == ticket) slot = &c;
    if (!slot) return fail("uva_net_collect_u8: ticket " + std::to_string(ticket) + " is not in flight");
    uva_net::PipeSlot& ps = *slot;
    HIP_TRY(hipSetDevice(n->device));
    if (ps.user_out) {
        const int per = (ps.out_rows + uva_net::PipeSlot::BANDS - 1) / uva_net::PipeSlot::BANDS;
        for (int k = 0; k < uva_net::PipeSlot::BANDS; ++k) {
            HIP_TRY(hipEventSynchronize(ps.ev_band[k]));
            const int r0 = std::min(ps.out_rows, k * per), r1 = std::min(ps.out_rows, r0 + per);
            if (ps.user_out_stride == ps.out_row) {
                if (r1 > r0) std::memcpy(ps.user_out + (size_t)r0 * ps.out_row, ps.h_out + (size_t)r0 * ps.out_row, (size_t)(r1 - r0) * ps.out_row);
            } else {
                for (int y = r0; y < r1; ++y)
                    std::memcpy(ps.user_out + (size_t)y * ps.user_out_stride, ps.h_out + (size_t)y * ps.out_row, ps.out_row);
            }
        }
        ps.busy = false;
        return 0;
    }
    HIP_TRY(hipEventSynchronize(ps.ev_d2h));
    if (ps.png_ws) {
        // the meta is here: fetch what the download did not cover (a frame that compressed worse than the one before)
        const size_t mb = png_meta_bytes(ps.png_h, ps.png_w), cap = png_workspace_bytes(ps.png_h, ps.png_w) - mb;
        const size_t total = png_total_bytes(ps.png_ws, ps.png_h, ps.png_w);
        if (total > cap) { ps.busy = false; ps.png_ws = nullptr; return fail("PNG encoder: block sizes out of range"); }
        if (total > ps.png_sent)
            HIP_TRY(hipMemcpy(ps.png_ws + mb + ps.png_sent, png_packed(ps.d_png, ps.png_h, ps.png_w) + ps.png_sent, total - ps.png_sent,
                              hipMemcpyDeviceToHost));
        n->png_guess = total;
        ps.png_ws = nullptr;
    }
    ps.busy = false;
    return 0;
}

int uva_net_process_u8(uva_net* n, const uint8_t* in, int h, int w, size_t in_stride, uint8_t* out,
                       size_t out_stride, int tile_size, int border)
{
    const long long t = uva_net_submit_u8(n, in, h, w, in_stride, out, out_stride, tile_size, border);
    if (t < 0) return 1;
    if (uva_net_collect_u8(n, t)) return 1;
    resolve_events(n);
    return 0;
}

int uva_net_extract_f32(uva_net* n, const float* in_chw, int h, int w, float* out_chw)
{
    if (check_dims(n, h, w)) return 1;
    if (!in_chw || !out_chw) return fail("null Mat pointer");
    if (ensure_device(n)) return 1;
    const int s = uva_net_scale(n);
    const size_t in_bytes = (size_t)3 * h * w * 4, out_bytes = (size_t)3 * h * s * w * s * 4;
    if (grow_dev(&n->d_fin, &n->d_fin_cap, in_bytes) || grow_dev(&n->d_fout, &n->d_fout_cap, out_bytes)) return 1;
    if (n->generic) {
        HIP_TRY(hipMemcpyAsync(n->d_fin, in_chw, in_bytes, hipMemcpyHostToDevice, n->stream));
        if (generic_run_plane(n, true, n->d_fin, 0, 0, 0, h, w, n->d_fout, 0, 0, h, 0, w)) return 1;
        HIP_TRY(hipMemcpyAsync(out_chw, n->d_fout, out_bytes, hipMemcpyDeviceToHost, n->stream));
        return uva_net_synchronize(n);
    }
    Workspace* ws = nullptr;
    if (get_workspace(n, h, w, 0, 0, &ws)) return 1;
    HIP_TRY(hipMemcpyAsync(n->d_fin, in_chw, in_bytes, hipMemcpyHostToDevice, n->stream));
    n->last.valid = true; n->last.ws = ws; n->last.f32 = true; n->last.src = n->d_fin; n->last.src_stride = 0;
    if (run_graph(n, ws, true, n->d_fin, 0, n->d_fout, 0, -1)) return 1;
    HIP_TRY(hipMemcpyAsync(out_chw, n->d_fout, out_bytes, hipMemcpyDeviceToHost, n->stream));
    return uva_net_synchronize(n);
}

int uva_net_debug_read_activation(uva_net* n, int conv_idx, float* out_chw, int h, int w)
{
    if (!n || !out_chw) return fail("null argument");
    if (!n->dev_ready || !n->last.valid) return fail("no previous call to replay");
    Workspace* ws = n->last.ws;
    const int nconv = (int)n->g.convs.size();
    if (ws->planes.size() != 1 || ws->h != h || ws->w != w) return fail("debug read needs an untiled call of the same size");
    if (conv_idx < 0 || conv_idx > nconv - 2) return fail("conv_idx out of range");
    HIP_TRY(hipSetDevice(n->device));
    if (run_graph(n, ws, n->last.f32, n->last.src, n->last.src_stride, nullptr, 0, conv_idx)) return 1;
    const PlaneDesc& p = ws->planes[0];
    const int nf = n->g.nf;
    const size_t rows = (size_t)p.nty * TH + 2;
    std::vector<uint16_t> hbuf(rows * p.pitch * nf);
    HIP_TRY(hipMemcpyAsync(hbuf.data(), ws->act[n->last_act_buf], hbuf.size() * 2, hipMemcpyDeviceToHost, n->stream));
    HIP_TRY(hipStreamSynchronize(n->stream));
    for (int c = 0; c < nf; ++c)
        for (int y = 0; y < h; ++y)
            for (int x = 0; x < w; ++x)
                out_chw[((size_t)c * h + y) * w + x] =
                    f16_bits_to_f32(hbuf[((size_t)(y + 1) * p.pitch + (x + 1)) * nf + c]);
    return 0;
}

int uva_net_set_profiling(uva_net* n, int enable)
{
    if (!n) return fail("null net");
    if (uva_net_synchronize(n)) return 1;
    n->prof = enable != 0;
    for (int k = 0; k < 3; ++k) { n->launches[k] = 0; n->total_ms[k] = 0; }
    return 0;
}

int uva_net_kernel_stats(uva_net* n, int kind, long long* launches, double* total_ms)
{
    if (!n || kind < 0 || kind > 2) return fail("bad argument");
    if (launches) *launches = n->launches[kind];
    if (total_ms) *total_ms = n->total_ms[kind];
    return 0;
}

#ifdef UVA_INSTRUMENT
// debug: one trunk-layer launch on the last call's workspace with in-kernel s_memtime stamps
// (block 0, wave 0): out[8*i + {0,1,2,3,4}] = tile i {start, k-loop done, barrier passed, epilogue done,
// epilogue staging written}
int uva_net_debug_trunk_stamps(uva_net* n, unsigned long long* out, int max_tiles, int* tiles, int ablate,
                               float* kernel_ms)
{
    if (!n || !out || !n->dev_ready || !n->last.valid) return fail("no previous call to replay");
    Workspace* ws = n->last.ws;
    HIP_TRY(hipSetDevice(n->device));
    unsigned long long* d = nullptr;
    const size_t bytes = (size_t)max_tiles * 8 * sizeof(unsigned long long);
    HIP_TRY(hipMalloc((void**)&d, bytes));
    HIP_TRY(hipMemsetAsync(d, 0, bytes, n->stream));
    ConvArgs ca;
    std::memset(&ca, 0, sizeof ca);
    ca.planes = ws->d_planes;
    ca.nplanes = (int)ws->planes.size();
    ca.ntiles = ws->ntiles;
    ca.tiles_per_xcd = (ws->ntiles + 7) / 8;
    ca.in_act = ws->act[0];
    ca.out_act = ws->act[1];
    ca.wpk = n->layers[1].wpk;
    ca.bias = n->layers[1].bias;
    ca.slope = n->layers[1].slope;
    ca.dbg = d;
    ca.sink = n->d_sink;
    const bool split = n->g.nf == 64;
    const int grid = std::max(8, (n->ncu / 8) * 8);
    const int g8 = (split ? 2 : 1) * grid / 8;
    const int txcd = ((split ? ws->ntiles4 : ws->ntiles) + 7) / 8;
    const int per_block = (txcd + g8 - 1) / g8;
    if (per_block > max_tiles) { (void)hipFree(d); return fail("max_tiles too small"); }
    hipEvent_t e0, e1;
    HIP_TRY(hipEventCreate(&e0));
    HIP_TRY(hipEventCreate(&e1));
    if (ablate == 3) {
        // the u8 tail kernel of the last host-route call instead of a trunk layer
        if (n->last.f32 || !n->last.dst) { (void)hipFree(d); return fail("tail stamps need a previous uva_net_process_u8 call"); }
        const int nconv = (int)n->g.convs.size();
        ca.in_act = ws->act[n->last_act_buf];
        ca.out_act = nullptr;
        ca.wpk = n->layers[nconv - 1].wpk;
        ca.bias = n->layers[nconv - 1].bias;
        ca.slope = nullptr;
        ca.src_u8 = (const uint8_t*)n->last.src;
        ca.src_stride = n->last.src_stride;
        ca.dst_u8 = (uint8_t*)n->last.dst;
        ca.dst_stride = n->last.dst_stride;
        int rc3 = launch_tail_u8(n, ws, ca);
        ca.dbg = nullptr;
        HIP_TRY(hipEventRecord(e0, n->stream));
        for (int r = 0; r < 10 && !rc3; ++r) rc3 = launch_tail_u8(n, ws, ca);
        HIP_TRY(hipEventRecord(e1, n->stream));
        ca.dbg = d;
        HIP_TRY(hipMemsetAsync(d, 0, bytes, n->stream));
        if (!rc3) rc3 = launch_tail_u8(n, ws, ca);
        if (!rc3) {
            HIP_TRY(hipMemcpyAsync(out, d, bytes, hipMemcpyDeviceToHost, n->stream));
            HIP_TRY(hipStreamSynchronize(n->stream));
            float ms = 0;
            HIP_TRY(hipEventElapsedTime(&ms, e0, e1));
            if (kernel_ms) *kernel_ms = ms / 10;
        }
        (void)hipEventDestroy(e0);
        (void)hipEventDestroy(e1);
        (void)hipFree(d);
        const int grid3 = std::max(8, (n->ncu / 8) * 8);
        const bool pingpong_tail = n->g.nf == 64 && (n->g.scale == 2 || n->g.scale == 4);   // 4-row tiles, 2 groups
        if (tiles) *tiles = pingpong_tail ? per_block : (ca.tiles_per_xcd + grid3 / 8 - 1) / (grid3 / 8);
        return rc3;
    }
    if (ablate == 7) {
        // sub10_kernel (the whole 1x net): out[(step*12 + wave)*4 + {0 step start, 1 MFMAs done, 2 at the barrier}] of
        // workgroup 0 (max_tiles*8 words must hold 40 per step); *tiles = steps
        if (n->last.f32 || !n->last.dst || ws->planes.size() != 1) { (void)hipFree(d); return fail("sub10 stamps need a previous whole-frame uva_net_process_u8 call"); }
        int rc7 = launch_sub10(n, ws, n->last.src, n->last.src_stride, n->last.dst, n->last.dst_stride);
        if (rc7 == 0 && (size_t)(ws->max_rows10[0] + S10_DRAIN + 2) * 4 * S10_NW > (size_t)max_tiles * 8) { (void)hipEventDestroy(e0); (void)hipEventDestroy(e1); (void)hipFree(d); return fail("max_tiles too small"); }
        HIP_TRY(hipEventRecord(e0, n->stream));
        for (int r = 0; r < 50 && !rc7; ++r) rc7 = launch_sub10(n, ws, n->last.src, n->last.src_stride, n->last.dst, n->last.dst_stride);
        HIP_TRY(hipEventRecord(e1, n->stream));
        if (!rc7) rc7 = launch_sub10(n, ws, n->last.src, n->last.src_stride, n->last.dst, n->last.dst_stride, d);
        if (!rc7) {
            HIP_TRY(hipMemcpyAsync(out, d, bytes, hipMemcpyDeviceToHost, n->stream));
            HIP_TRY(hipStreamSynchronize(n->stream));
            float ms = 0;
            HIP_TRY(hipEventElapsedTime(&ms, e0, e1));
            if (kernel_ms) *kernel_ms = ms / 50;
        }
        (void)hipEventDestroy(e0);
        (void)hipEventDestroy(e1);
        (void)hipFree(d);
        if (tiles) *tiles = ws->max_rows10[0] + S10_DRAIN;
        return rc7 ? (rc7 == 2 ? fail("frame too large for sub10_kernel") : 1) : 0;
    }
    if (ablate == 9 || ablate == 10) {
        // sub5_kernel, part 0 / part 1 (the 1x net as two launches of five layers): out[(step*12 + wave)*4 + {0 step start, 1 MFMAs
        // done (front wave), 2 at the barrier}] of workgroup 0; *tiles = steps; kernel_ms = both launches
        if (n->last.f32 || !n->last.dst || ws->planes.size() != 1) { (void)hipFree(d); return fail("sub5 stamps need a previous whole-frame uva_net_process_u8 call"); }
        int rc9 = launch_sub5(n, ws, n->last.src, n->last.src_stride, n->last.dst, n->last.dst_stride);
        if (rc9 == 0 && (size_t)(ws->max_rows5 + 16) * 4 * 12 > (size_t)max_tiles * 8) { (void)hipEventDestroy(e0); (void)hipEventDestroy(e1); (void)hipFree(d); return fail("max_tiles too small"); }
        HIP_TRY(hipEventRecord(e0, n->stream));
        for (int r = 0; r < 50 && !rc9; ++r) rc9 = launch_sub5(n, ws, n->last.src, n->last.src_stride, n->last.dst, n->last.dst_stride);
        HIP_TRY(hipEventRecord(e1, n->stream));
        if (!rc9) rc9 = launch_sub5(n, ws, n->last.src, n->last.src_stride, n->last.dst, n->last.dst_stride, d, ablate - 9);
        if (!rc9) {
            HIP_TRY(hipMemcpyAsync(out, d, bytes, hipMemcpyDeviceToHost, n->stream));
            HIP_TRY(hipStreamSynchronize(n->stream));
            float ms = 0;
            HIP_TRY(hipEventElapsedTime(&ms, e0, e1));
            if (kernel_ms) *kernel_ms = ms / 50;
        }
        (void)hipEventDestroy(e0);
        (void)hipEventDestroy(e1);
        (void)hipFree(d);
        if (tiles) *tiles = ws->max_rows5 + 14;
        return rc9 ? (rc9 == 2 ? fail("frame too large for sub5_kernel") : 1) : 0;
    }
    if (ablate == 6) {
        // pair24_kernel (two 24-feature trunk layers): out[8*it + {0 top, 1 tile landed, 2 stage A done, 3 intermediate
        // complete, 4 stage B k-loop done, 5 stores issued}] of workgroup 0 / wave 0
        if (n->g.nf != 24) { (void)hipFree(d); return fail("pair24 stamps need the 24-feature net"); }
        ca.wpk2 = n->layers[2].wpk; ca.bias2 = n->layers[2].bias; ca.slope2 = n->layers[2].slope;
        ca.ntiles = ws->ntiles; ca.tiles_per_xcd = (ws->ntiles + 7) / 8;
        ca.dbg = nullptr;
        int rc6 = launch_pair24(n, ca);
        HIP_TRY(hipEventRecord(e0, n->stream));
        for (int r = 0; r < 50 && !rc6; ++r) rc6 = launch_pair24(n, ca);
        HIP_TRY(hipEventRecord(e1, n->stream));
        ca.dbg = d;
        if (!rc6) rc6 = launch_pair24(n, ca);
        if (!rc6) {
            HIP_TRY(hipMemcpyAsync(out, d, bytes, hipMemcpyDeviceToHost, n->stream));
            HIP_TRY(hipStreamSynchronize(n->stream));
            float ms = 0;
            HIP_TRY(hipEventElapsedTime(&ms, e0, e1));
            if (kernel_ms) *kernel_ms = ms / 50;
        }
        (void)hipEventDestroy(e0);
        (void)hipEventDestroy(e1);
        (void)hipFree(d);
        const int grid6 = std::max(8, (n->ncu / 8) * 8) * 2;
        if (tiles) *tiles = (ca.tiles_per_xcd + grid6 / 8 - 1) / (grid6 / 8);
        return rc6;
    }
    if (ablate == 8) {
        // trunkw_kernel (the fused pair as Winograd F(2,3)): stamps of workgroup 0, both groups (out[16*it + 8*group + k], entry
        // at out[16*niter]); only an instrumented build (-DUVA_INSTRUMENT) writes them
        if (!ws->d_stepsw || !n->layers[1].wpk_w) { (void)hipFree(d); return fail("no Winograd fused-pair schedule for this net"); }
        TrunkwArgs wa;
        std::memset(&wa, 0, sizeof wa);
        wa.in_act = ws->act_base[0];
        wa.out_act = ws->act_base[1];
        wa.steps = ws->d_stepsw; wa.nsteps = ws->d_nstepsw; wa.max_steps = ws->max_stepsw; wa.sink = n->d_sink;
        if (2 * (ws->max_stepsw + 3) + 1 > max_tiles) { (void)hipFree(d); return fail("max_tiles too small"); }
        int rc8 = launch_trunkw(n, ws, wa, 1);
        HIP_TRY(hipEventRecord(e0, n->stream));
        for (int r = 0; r < 50 && !rc8; ++r) rc8 = launch_trunkw(n, ws, wa, 1);
        HIP_TRY(hipEventRecord(e1, n->stream));
        wa.dbg = d;
        if (!rc8) rc8 = launch_trunkw(n, ws, wa, 1);
        if (!rc8) {
            HIP_TRY(hipMemcpyAsync(out, d, bytes, hipMemcpyDeviceToHost, n->stream));
            HIP_TRY(hipStreamSynchronize(n->stream));
            float ms = 0;
            HIP_TRY(hipEventElapsedTime(&ms, e0, e1));
            if (kernel_ms) *kernel_ms = ms / 50;
        }
        (void)hipEventDestroy(e0);
        (void)hipEventDestroy(e1);
        (void)hipFree(d);
        int ns0 = 0;
        if (!rc8) HIP_TRY(hipMemcpy(&ns0, ws->d_nstepsw, sizeof(int), hipMemcpyDeviceToHost));
        if (tiles) *tiles = ns0 + 2;
        return rc8;
    }
    if (ablate == 5) {
        // the fused pair kernel: stamps of workgroup 0, both groups (out[16*it + 8*group + k], entry at out[16*niter])
        if (!ws->d_steps2) { (void)hipFree(d); return fail("no fused-pair schedule for this net"); }
        Trunk2Args ta;
        std::memset(&ta, 0, sizeof ta);
        ta.in_act = ws->act_base[0];
        ta.out_act = ws->act_base[1];
        for (int k = 0; k < 2; ++k) { ta.wpk[k] = n->layers[1 + k].wpk; ta.bias[k] = n->layers[1 + k].bias; ta.slope[k] = n->layers[1 + k].slope; }
        ta.steps = ws->d_steps2; ta.nsteps = ws->d_nsteps2; ta.max_steps = ws->max_steps2; ta.sink = n->d_sink;
        if (2 * (ws->max_steps2 + 3) + 1 > max_tiles) { (void)hipFree(d); return fail("max_tiles too small"); }
        int rc5 = launch_trunk2(n, ws, ta);
        HIP_TRY(hipEventRecord(e0, n->stream));
        for (int r = 0; r < 50 && !rc5; ++r) rc5 = launch_trunk2(n, ws, ta);
        HIP_TRY(hipEventRecord(e1, n->stream));
        ta.dbg = d;
        if (!rc5) rc5 = launch_trunk2(n, ws, ta);
        if (!rc5) {
            HIP_TRY(hipMemcpyAsync(out, d, bytes, hipMemcpyDeviceToHost, n->stream));
            HIP_TRY(hipStreamSynchronize(n->stream));
            float ms = 0;
            HIP_TRY(hipEventElapsedTime(&ms, e0, e1));
            if (kernel_ms) *kernel_ms = ms / 50;
        }
        (void)hipEventDestroy(e0);
        (void)hipEventDestroy(e1);
        (void)hipFree(d);
        std::vector<int> ns(ws->grid2);
        HIP_TRY(hipMemcpy(ns.data(), ws->d_nsteps2, ns.size() * sizeof(int), hipMemcpyDeviceToHost));
        if (tiles) *tiles = ns[0] + 2;
        return rc5;
    }
    // bits 8.. of `ablate`: hundreds of timed repetitions (sustained, power-limited state) instead of 10
    const int reps = (ablate >> 8) > 0 ? (ablate >> 8) * 100 : 10;
    ablate &= 0xff;
    int rc = launch_trunk(n, ws, ca, ablate);   // warm
    ca.dbg = nullptr;
    HIP_TRY(hipEventRecord(e0, n->stream));
    for (int r = 0; r < reps && !rc; ++r) rc = launch_trunk(n, ws, ca, ablate);
    HIP_TRY(hipEventRecord(e1, n->stream));
    ca.dbg = d;
    HIP_TRY(hipMemsetAsync(d, 0, bytes, n->stream));
    if (!rc) rc = launch_trunk(n, ws, ca, ablate);
    if (!rc) {
        HIP_TRY(hipMemcpyAsync(out, d, bytes, hipMemcpyDeviceToHost, n->stream));
        HIP_TRY(hipStreamSynchronize(n->stream));
        float ms = 0;
        HIP_TRY(hipEventElapsedTime(&ms, e0, e1));
        if (kernel_ms) *kernel_ms = ms / reps;
    }
    (void)hipEventDestroy(e0);
    (void)hipEventDestroy(e1);
    (void)hipFree(d);
    if (tiles) *tiles = per_block;
    return rc;
}

// debug: rdb4_kernel's in-kernel stamps of the last launch (UVA_RDB_STAMPS=1): out[(step * 4 + wave) * 4 + k]
int uva_net_debug_rdb_stamps(uva_net* n, unsigned long long* out, int max_steps)
{
    if (!n || !out || !n->gd.rdb_dbg) return fail("no rdb4 stamps (UVA_RDB_STAMPS=1, a generic graph, one call)");
    HIP_TRY(hipSetDevice(n->device));
    HIP_TRY(hipStreamSynchronize(n->stream));
    HIP_TRY(hipMemcpy(out, n->gd.rdb_dbg, (size_t)std::min(max_steps, 1024) * 16 * 8, hipMemcpyDeviceToHost));
    return 0;
}

#endif  // UVA_INSTRUMENT

// test hook: the packed MFMA weight image of convolution #conv_idx (host side, no device needed)
int uva_net_debug_packed_weights(uva_net* n, int conv_idx, uint16_t* out, size_t out_halfs, size_t* needed)
{
    if (!n || !n->g.model_loaded) return fail("no model");
    if (conv_idx == -1 && n->g.nf != 64) return fail("conv_idx -1 (tail_kernel image) exists for the 64-feature nets only");
    if (conv_idx < -1 || conv_idx >= (int)n->g.convs.size()) return fail("conv_idx out of range");
    std::vector<uint16_t> pk;
    if (conv_idx == -1) pack_tail64(n->g.convs.back(), pk);          // tail_kernel's image of the last convolution
    else if (conv_idx == 0) pack_head(n->g.convs[0], pk, nullptr);
    else if (n->g.nf == 64 && conv_idx + 1 < (int)n->g.convs.size()) pack_trunk64(n->g.convs[conv_idx], pk);
    else pack_conv3x3(n->g.convs[conv_idx], n->g.nf, pk, nullptr, nullptr);
    if (needed) *needed = pk.size();
    if (out && out_halfs >= pk.size()) std::memcpy(out, pk.data(), pk.size() * 2);
    return 0;
}

int uva_debug_trunk2_schedule(int h, int w, int tile_size, int border, int grid, uint32_t* steps_words,
                              size_t capacity_words, size_t* needed_words, int* nsteps, int* stride,
                              long long* plane_info, int max_planes, int* nplanes, long long* guard_bytes)
{
    if (h <= 0 || w <= 0 || grid < 8 || grid % 8) return fail("bad argument");
    std::vector<PlaneDesc> planes;
    if (tile_size <= 0) { tile_size = 0; border = 0; }
    if (build_planes(h, w, tile_size, border, planes)) return 1;
    size_t pix = 0;
    int max_pitch = 0;
    layout_planes(planes, &pix, nullptr, nullptr);
    for (auto& p : planes) max_pitch = std::max(max_pitch, p.pitch);
    const size_t guard = (size_t)8 * max_pitch * 128;
    std::vector<Trunk2Step> steps;
    std::vector<int> ns;
    int max_steps = 0;
    if (build_trunk2_schedule(planes, grid, guard, steps, ns, &max_steps)) return 1;
    if (needed_words) *needed_words = steps.size() * 8;
    if (stride) *stride = max_steps + T2_PAD_STEPS;
    if (nplanes) *nplanes = (int)planes.size();
    if (guard_bytes) *guard_bytes = (long long)guard;
    if (plane_info)
        for (int i = 0; i < (int)planes.size() && i < max_planes; ++i) {
            plane_info[4 * i + 0] = planes[i].h; plane_info[4 * i + 1] = planes[i].w;
            plane_info[4 * i + 2] = planes[i].pitch; plane_info[4 * i + 3] = planes[i].act_off;
        }
    if (!steps_words || capacity_words < steps.size() * 8) return fail("steps buffer too small");
    std::memcpy(steps_words, steps.data(), steps.size() * sizeof(Trunk2Step));
    if (nsteps) std::copy(ns.begin(), ns.end(), nsteps);
    return 0;
}

int uva_debug_trunkw_schedule(int h, int w, int tile_size, int border, int grid, uint32_t* steps_words,
                              size_t capacity_words, size_t* needed_words, int* nsteps, int* stride,
                              long long* plane_info, int max_planes, int* nplanes, long long* guard_bytes)
{
    if (h <= 0 || w <= 0 || grid < 8 || grid % 8) return fail("bad argument");
    std::vector<PlaneDesc> planes;
    if (tile_size <= 0) { tile_size = 0; border = 0; }
    if (build_planes(h, w, tile_size, border, planes)) return 1;
    size_t pix = 0;
    int max_pitch = 0;
    layout_planes(planes, &pix, nullptr, nullptr);
    for (auto& p : planes) max_pitch = std::max(max_pitch, p.pitch);
    const size_t guard = (size_t)8 * max_pitch * 128;
    std::vector<Trunk2Step> steps;
    std::vector<int> ns;
    int max_steps = 0;
    if (build_trunkw_schedule(planes, grid, guard, steps, ns, &max_steps)) return 1;
    if (needed_words) *needed_words = steps.size() * 8;
    if (stride) *stride = max_steps + TW_PAD_STEPS;
    if (nplanes) *nplanes = (int)planes.size();
    if (guard_bytes) *guard_bytes = (long long)guard;
    if (plane_info)
        for (int i = 0; i < (int)planes.size() && i < max_planes; ++i) {
            plane_info[4 * i + 0] = planes[i].h; plane_info[4 * i + 1] = planes[i].w;
            plane_info[4 * i + 2] = planes[i].pitch; plane_info[4 * i + 3] = planes[i].act_off;
        }
    if (!steps_words || capacity_words < steps.size() * 8) return fail("steps buffer too small");
    std::memcpy(steps_words, steps.data(), steps.size() * sizeof(Trunk2Step));
    if (nsteps) std::copy(ns.begin(), ns.end(), nsteps);
    return 0;
}

int uva_debug_sub10_rows(int h, int w, int grid, uint32_t* rows_words, size_t capacity_words, size_t* needed_words,
                         int* nrows, int* stride)
{
    return uva_debug_sub10_rows_batch(h, w, 1, grid, rows_words, capacity_words, needed_words, nrows, stride);
}

int uva_debug_sub10_rows_batch(int h, int w, int frames, int grid, uint32_t* rows_words, size_t capacity_words, size_t* needed_words,
                               int* nrows, int* stride)
{
    if (h <= 0 || w <= 0 || grid < 8 || grid % 8 || frames < 1 || frames > S10_MAXB) return fail("bad argument");
    std::vector<uint4> rows;
    std::vector<int> nr;
    int max_rows = 0;
    const int rc = build_sub10_rows(h, w, frames, grid, rows, nr, &max_rows);
    if (rc == 2) return fail("frame too large for the fused 1x kernel's row table");
    if (rc) return 1;
    if (needed_words) *needed_words = rows.size() * 4;
    if (stride) *stride = max_rows;
    if (!rows_words || capacity_words < rows.size() * 4) return fail("rows buffer too small");
    std::memcpy(rows_words, rows.data(), rows.size() * sizeof(uint4));
    if (nrows) std::copy(nr.begin(), nr.end(), nrows);
    return 0;
}

int uva_debug_sub5_rows(int h, int w, int grid, uint32_t* rows_words, size_t capacity_words, size_t* needed_words, int* nrows, int* stride)
{
    if (h <= 0 || w <= 0 || grid <= 0 || grid % 8) return fail("bad argument");
    std::vector<uint4> rows;
    std::vector<int> nr;
    int max_rows = 0;
    const int rc = build_sub5_rows(h, w, grid, rows, nr, &max_rows);
    if (rc == 2) return fail("frame too large for sub5_kernel's row table");
    if (rc) return 1;
    if (needed_words) *needed_words = rows.size() * 4;
    if (stride) *stride = max_rows;
    if (!rows_words || capacity_words < rows.size() * 4) return fail("rows buffer too small");
    std::memcpy(rows_words, rows.data(), rows.size() * sizeof(uint4));
    if (nrows) std::copy(nr.begin(), nr.end(), nrows);
    return 0;
}

int uva_debug_generic_segments_planes(int kind, const int* dims, int nplanes, int grid, int32_t* out, size_t capacity_words,
                                      size_t* needed_words, int* seg_begin)
{
    if (!dims || nplanes <= 0 || nplanes > GEN_MAX_PLANES || grid <= 0) return fail("bad argument");
    for (int i = 0; i < 2 * nplanes; ++i)
        if (dims[i] <= 0) return fail("bad argument");
    const std::vector<int> d(dims, dims + 2 * nplanes);
    std::vector<int32_t> words;
    std::vector<int> sbeg;
    if (kind == 0) {
        std::vector<RdbSeg> segs;
        rdb_segments(d, grid, segs, sbeg);
        for (const RdbSeg& sg : segs) words.insert(words.end(), {sg.c0, sg.yb, sg.ye, sg.own0, sg.own1, sg.plane, 0, 0});
    } else if (kind == 1 || kind == 2) {
        std::vector<GSwSeg> segs;
        const int cols = kind == 1 ? sw_cols<1>() : sw_cols<2>();
        sw_segments(d, cols, grid, segs, sbeg);
        for (const GSwSeg& sg : segs) words.insert(words.end(), {sg.c0, sg.y0, sg.y1, sg.c0, sg.c0 + cols, sg.plane, 0, 0});
    } else {
        return fail("bad kind");
    }
    if (needed_words) *needed_words = words.size();
    if (!out || capacity_words < words.size()) return fail("segment buffer too small");
    std::memcpy(out, words.data(), words.size() * sizeof(int32_t));
    if (seg_begin) std::copy(sbeg.begin(), sbeg.end(), seg_begin);
    return 0;
}

int uva_debug_generic_batches(int h, int w, int tile_size, int border, long long batch_pixels, int32_t* out, size_t capacity_words,
                              size_t* needed_words)
{
    if (h <= 0 || w <= 0) return fail("bad argument");
    std::vector<PlaneDesc> planes;
    if (tile_size <= 0) { tile_size = 0; border = 0; }
    if (build_planes(h, w, tile_size, border, planes)) return 1;
    std::vector<int> batch_of;
    generic_plan_batches(planes, true, batch_pixels > 0 ? batch_pixels : generic_batch_pixels(), batch_of);
    if (needed_words) *needed_words = planes.size() * 4;
    if (!out || capacity_words < planes.size() * 4) return fail("batch buffer too small");
    for (size_t i = 0; i < planes.size(); ++i) {
        out[4 * i] = planes[i].h; out[4 * i + 1] = planes[i].w; out[4 * i + 2] = batch_of[i]; out[4 * i + 3] = generic_plane_class(planes[i].w);
    }
    return 0;
}

int uva_debug_generic_segments(int kind, int h, int w, int grid, int32_t* out, size_t capacity_words, size_t* needed_words, int* seg_begin)
{
    const int dims[2] = {h, w};
    return uva_debug_generic_segments_planes(kind, dims, 1, grid, out, capacity_words, needed_words, seg_begin);
}

}  // extern "C"
