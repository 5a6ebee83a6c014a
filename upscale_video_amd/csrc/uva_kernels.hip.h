// uva_kernels.hip.h -- CDNA4 (gfx950) kernels for the SRVGGNetCompact graphs that
// davlee1972/upscale_video runs through ncnn-vulkan (reference graphs:
// models/2x_Compact_Pretrain.param:3-42, models/4x_Compact_Pretrain.param:3-42,
// models/1x_HurrDeblur_SubCompact_nf24-nc8_244k_net_g.param:3-26; call sites
// upscale/upscale_processing.py:263-288 and :430-477).
//
// Data layout in HBM
//   activations : fp16 NHWC, one zero-bordered plane per sub-image: pixel (y,x) of a plane lives
//                 at row y+1, column x+1 of a (nty*TH+2) x (ntx*TW+2) pixel array.  The border and
//                 everything outside the image is zero and is never written, so the 3x3 zero
//                 padding of ncnn's Convolution (pad=1) costs no bounds checks: every halo load
//                 is in range.
//   weights     : fp16, pre-packed per layer in MFMA A-operand order [k-step][m-frag][lane][8].
//   frame I/O   : u8 HWC BGR exactly as cv2.imread / cv2.imwrite hold it, or f32 planar CHW
//                 exactly as ncnn::Mat holds it.
//
// Kernel structure (conv3x3_kernel): one persistent 4-wave workgroup per CU.  Every wave keeps the
// layer's whole weight matrix (Cout x 9*Cin, fp16) in its 512-entry VGPR/AGPR file for the
// lifetime of the kernel, so the MFMA A operand never touches LDS or L2 again.  Work tiles are
// 8 rows x 32 columns of output pixels; the 10 x 34 pixel input halo tile is streamed
// HBM/L2 -> LDS by LDS-DMA (global_load_lds_dwordx4, no VGPR round trip), double buffered so the
// DMA of tile t+1 runs under the MFMAs of tile t.  v_mfma_f32_32x32x16_f16: A = weights
// (rows = output channels), B = 32 consecutive pixels of one image row (cols), K = 16 input
// channels of one tap, fp32 accumulate.  Each B fragment read from LDS feeds Cout/32 MFMAs.
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>

namespace uva {

typedef _Float16 half8 __attribute__((ext_vector_type(8)));
typedef _Float16 half4 __attribute__((ext_vector_type(4)));
typedef float f32x16 __attribute__((ext_vector_type(16)));
typedef float f32x4 __attribute__((ext_vector_type(4)));

constexpr int TH = 8;           // work-tile rows
constexpr int TW = 32;          // work-tile columns (= one MFMA N fragment)
constexpr int PH = TH + 2;      // halo tile rows
constexpr int PW = TW + 2;      // halo tile columns
constexpr int NPIX = PH * PW;   // 340 pixels per halo tile
constexpr int MAX_PLANES = 64;

// One independent sub-image.  The reference cuts a frame into <=980x980 tiles
// (upscale_processing.py:395-434) and feeds each through the net on its own; each such tile is a
// "plane" here, and all planes of a frame go through every layer in one launch.
struct PlaneDesc {
    int h, w;               // plane size in input pixels
    int nty, ntx;           // work tiles
    int tile_begin;         // first global work-tile index of this plane
    int pitch;              // activation row pitch in pixels (ntx*TW + 2)
    long long act_off;      // pixel offset of the plane's padded array inside the activation buffer
    int src_y0, src_x0;     // plane origin inside the source frame
    int core_y0, core_y1;   // plane-local rows whose output is written (border cropped, :464-477)
    int core_x0, core_x1;   // plane-local columns whose output is written
    int pad0, pad1;
};
static_assert(sizeof(PlaneDesc) == 64, "PlaneDesc layout");

template <int NF>
struct Geo {
    static constexpr int SPP = NF / 8;                       // 16-byte channel octets per pixel
    static constexpr int LSPP = (NF == 64) ? 9 : SPP;        // LDS slots per pixel (64ch: +1 pad slot
                                                             //  -> 144 B stride, conflict-free b128)
    static constexpr int LPIXB = LSPP * 16;                  // LDS bytes per pixel
    static constexpr int PIXB = NF * 2;                      // HBM bytes per pixel
    static constexpr int KO = 9 * SPP;                       // K octets (tap-major, then channel octet)
    static constexpr int KS = (KO + 1) / 2;                  // MFMA k-steps of 16
    static constexpr int NSLOT = NPIX * LSPP;
    static constexpr int NCHUNK = (NSLOT + 63) / 64;         // 1-KiB LDS-DMA pieces per halo tile
    static constexpr int BUFB = NCHUNK * 1024;
};

constexpr int PARAM_LDS = 768;   // bias[64] + slope[64] + PReLU med3 selector[64] floats

struct ConvArgs {
    const PlaneDesc* planes;
    int nplanes;
    int ntiles;
    int tiles_per_xcd;
    const _Float16* in_act;
    _Float16* out_act;            // trunk
    const half8* wpk;             // packed weights [KS][MF][64]
    const float* bias;            // [MF*32], zero padded
    const float* slope;           // [MF*32] (trunk only)
    const uint8_t* src_u8;        // tail: residual source frame, u8 HWC
    size_t src_stride;
    const float* src_f32;         // tail (f32 mode): residual source, planar [3][h][w]
    uint8_t* dst_u8;
    size_t dst_stride;
    float* dst_f32;               // planar [3][h*R][w*R]
};

struct HeadArgs {
    const PlaneDesc* planes;
    int nplanes;
    int ntiles;
    const uint8_t* src_u8;
    size_t src_stride;
    const float* src_f32;         // planar [3][h][w] (single plane)
    _Float16* out_act;
    const half8* wpk;             // [3][MF][64]
    const float* bias;
    const float* slope;
    float in_scale;               // 1/255 for u8 sources (applied to the fp32 accumulator), else 1
};

__device__ __forceinline__ unsigned lds_offset(const void* p)
{
    return (unsigned)(size_t)(const __attribute__((address_space(3))) char*)p;
}

// One LDS-DMA piece: 64 lanes x 16 B from per-lane global addresses to lds_dst + lane*16.
// Invisible to hipcc's s_waitcnt bookkeeping by design (it would otherwise drain the DMA before
// every LDS read); completion is waited for explicitly with vmcnt(0) in tile_barrier().  No
// "memory" clobber: the pieces are issued between the MFMAs of the previous tile and must not
// fence its LDS reads; ordering against the buffers comes from tile_barrier() alone.
__device__ __forceinline__ void glds16(const void* gsrc, unsigned lds_dst)
{
    unsigned keep;
    asm volatile(
        "s_mov_b32 %0, m0\n\t"
        "s_mov_b32 m0, %2\n\t"
        "s_nop 0\n\t"
        "global_load_lds_dwordx4 %1, off\n\t"
        "s_mov_b32 m0, %0"
        : "=&s"(keep)
        : "v"(gsrc), "s"(lds_dst));
}

__device__ __forceinline__ void tile_barrier()
{
    asm volatile("s_waitcnt vmcnt(0) lgkmcnt(0)\n\ts_barrier" ::: "memory");
}

struct TileId { int plane, ty, tx; };

// Which plane owns work tile t: planes are sorted by tile_begin, lane i looks at plane i.
// 'tile_begin_of(i)' reads from LDS (persistent kernels) or global memory (head kernel).
template <typename F>
__device__ __forceinline__ int find_plane(F tile_begin_of, int nplanes, int t, int lane)
{
    const int tb = lane < nplanes ? tile_begin_of(lane) : 0x7fffffff;
    const unsigned long long m = __ballot(t >= tb);
    return __builtin_popcountll(m) - 1;
}

// The plane table lives in LDS inside the persistent kernels: per-tile lookups must not be VMEM
// loads, whose vmcnt wait would also wait for the previous tile's output stores.
struct PlaneTable {
    const PlaneDesc* pl;   // LDS copy
    const int* tile_begin; // LDS, [MAX_PLANES]
    int nplanes;
    __device__ __forceinline__ TileId decode(int t, int lane) const
    {
        const int* tb = tile_begin;
        TileId id;
        id.plane = __builtin_amdgcn_readfirstlane(find_plane([tb](int i) { return tb[i]; }, nplanes, t, lane));
        const int ntx = __builtin_amdgcn_readfirstlane(pl[id.plane].ntx);
        const int local = t - __builtin_amdgcn_readfirstlane(pl[id.plane].tile_begin);
        id.ty = local / ntx;
        id.tx = local - id.ty * ntx;
        return id;
    }
};

constexpr int PLANE_LDS = MAX_PLANES * 64 + MAX_PLANES * 4;

// LDS-DMA piece i (of this wave) of the (TH+2)x(TW+2) halo tile: piece c = 4*i + wave covers LDS
// slots [64c, 64c+64); slot q is pixel q / LSPP, octet q % LSPP.  The lane's source position is
// loop-invariant and kept packed in one register: (halo row << 16) | byte offset inside the row.
template <int NF>
__device__ __forceinline__ int dma_piece_const(int i, int wave, int lane)
{
    using G = Geo<NF>;
    static_assert(G::NCHUNK % 4 == 0, "every wave issues the same number of pieces");
    const int q = (i * 4 + wave) * 64 + lane;
    int p = q / G::LSPP;
    int s = q - p * G::LSPP;
    if (s >= G::SPP) s = G::SPP - 1;      // pad slot: re-fetch the neighbouring octet
    if (p >= NPIX) p = NPIX - 1;          // tail of the last piece: any valid address
    const int r = p / PW;
    const int cc = p - r * PW;
    return (r << 16) | (cc * G::PIXB + s * 16);
}

template <int NF>
__device__ __forceinline__ void issue_dma_piece(const char* tile_base, int pitch_bytes, unsigned lds_buf, int i,
                                                int wave, int pc)
{
    const unsigned off = (unsigned)(pc >> 16) * (unsigned)pitch_bytes + (unsigned)(pc & 0xffff);
    glds16(tile_base + off, lds_buf + (i * 4 + wave) * 1024);
}

template <int NF>
__device__ __forceinline__ const char* halo_tile_base(const _Float16* act, const PlaneDesc& pl, int ty, int tx)
{
    return (const char*)act +
           ((size_t)pl.act_off + (size_t)(ty * TH) * pl.pitch + (size_t)tx * TW) * Geo<NF>::PIXB;
}

__device__ __forceinline__ int opaque(int v)
{
    asm volatile("" : "+v"(v));
    return v;
}

typedef _Float16 half2v __attribute__((ext_vector_type(2)));
typedef float f32x2 __attribute__((ext_vector_type(2)));

// Trunk epilogue shared by the head and trunk kernels: per-channel PReLU (ncnn prelu.cpp:
// x < 0 ? x*slope[c] : x), fp32 -> fp16 RNE, 8-byte stores into the zero-bordered NHWC plane.
// PReLU without a compare/select pair: x < 0 ? x*s : x  ==  s <= 1 ? max(x, x*s) : min(x, x*s)
// (exactly, rounding is monotonic), and both are med3(x, x*s, +-inf); prm_lds holds the slopes at
// [0,64) and the matching +-inf at [64,128).
template <int NF, int MF>
__device__ __forceinline__ void store_trunk(const f32x16 (&acc)[MF], const float* prm_lds, _Float16* out_act,
                                            const PlaneDesc& pl, int y, int x, int half)
{
    if (y >= pl.h || x >= pl.w) return;
    char* dst = (char*)out_act + ((size_t)pl.act_off + (size_t)(y + 1) * pl.pitch + (x + 1)) * (NF * 2);
#pragma unroll
    for (int m = 0; m < MF; ++m) {
#pragma unroll
        for (int g = 0; g < 4; ++g) {
            if (32 * m + 8 * g >= NF) continue;   // NF is a multiple of 8: groups are all-or-nothing
            const int cb = 32 * m + 8 * g + 4 * half;
            const f32x4 s4 = *(const f32x4*)(prm_lds + cb);
            const f32x4 i4 = *(const f32x4*)(prm_lds + 64 + cb);
            f32x4 v;
#pragma unroll
            for (int j = 0; j < 4; ++j) {
                const float x0 = acc[m][4 * g + j];
                v[j] = __builtin_amdgcn_fmed3f(x0, x0 * s4[j], i4[j]);
            }
            const half2v lo = __builtin_convertvector(f32x2{v[0], v[1]}, half2v);
            const half2v hi = __builtin_convertvector(f32x2{v[2], v[3]}, half2v);
            uint2 o;
            o.x = __builtin_bit_cast(unsigned, lo);
            o.y = __builtin_bit_cast(unsigned, hi);
            *(uint2*)(dst + cb * 2) = o;
        }
    }
}

// ----------------------------------------------------------------------------------------------
// conv3x3_kernel<NF, MODE, R>
//   MODE 0: trunk layer NF -> NF, bias + PReLU, fp16 NHWC out.
//   MODE 1: tail layer NF -> 3*R*R, bias, PixelShuffle(R) + nearest-upsampled normalised input
//           (ncnn pixelshuffle.cpp / interp.cpp resize_type 1 / binaryop.cpp ADD), then the
//           reference's *255 and cv2 convertTo(CV_8U) (round-half-even, saturate), u8 HWC out,
//           only the plane's core region (process_tile's crop, upscale_processing.py:464-477).
//   MODE 2: same tail arithmetic up to the add, f32 planar CHW out (np.array(mat_out), :281/:453).
// ----------------------------------------------------------------------------------------------
template <int NF, int MODE, int R>
__global__ __launch_bounds__(256, 1) void conv3x3_kernel(ConvArgs a)
{
    using G = Geo<NF>;
    constexpr int COUT = (MODE == 0) ? NF : 3 * R * R;
    constexpr int MF = (COUT + 31) / 32;
    constexpr int KS = G::KS;
    constexpr int STAGE_ROWB = TW * R * 3;             // bytes per staged output row
    constexpr int STAGEB = 2 * R * STAGE_ROWB;         // per wave: 2 tile rows -> 2R output rows

    extern __shared__ __attribute__((aligned(16))) char smem[];
    const unsigned lds0 = lds_offset(smem);
    float* bias_lds = (float*)(smem + 2 * G::BUFB);
    float* slope_lds = bias_lds + 64;
    PlaneDesc* planes_lds = (PlaneDesc*)(smem + 2 * G::BUFB + PARAM_LDS);
    int* tile_begin_lds = (int*)(smem + 2 * G::BUFB + PARAM_LDS + MAX_PLANES * 64);
    uint8_t* stage_all = (uint8_t*)(smem + 2 * G::BUFB + PARAM_LDS + PLANE_LDS);

    const int wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
    const int lane = threadIdx.x & 63;
    const int half = lane >> 5;
    const int px = lane & 31;

    // persistent schedule: XCD x owns the contiguous tile range [x*tiles_per_xcd, ...), the
    // blocks of one XCD (blockIdx % 8 == x) walk it 'gridDim/8' tiles at a time, so tiles in
    // flight on one XCD are neighbours and share halo lines in that XCD's L2.
    const int g8 = gridDim.x >> 3;
    const int xcd = blockIdx.x & 7;
    const int slot = blockIdx.x >> 3;
    const int t_first = xcd * a.tiles_per_xcd + slot;
    const int t_lim = min((xcd + 1) * a.tiles_per_xcd, a.ntiles);
    if (t_first >= t_lim) return;
    const int niter = (t_lim - t_first + g8 - 1) / g8;

    if (threadIdx.x < 64) {
        bias_lds[threadIdx.x] = threadIdx.x < MF * 32 ? a.bias[threadIdx.x] : 0.f;
        const float sl = (MODE == 0 && threadIdx.x < MF * 32) ? a.slope[threadIdx.x] : 0.f;
        slope_lds[threadIdx.x] = sl;
        slope_lds[64 + threadIdx.x] = sl <= 1.f ? __builtin_inff() : -__builtin_inff();
        tile_begin_lds[threadIdx.x] = threadIdx.x < a.nplanes ? a.planes[threadIdx.x].tile_begin : 0x7fffffff;
    }
    for (int i = threadIdx.x; i < a.nplanes * 16; i += 256) ((int*)planes_lds)[i] = ((const int*)a.planes)[i];

    // the layer's weights, resident in registers for the whole kernel
    half8 w[KS][MF];
#pragma unroll
    for (int ks = 0; ks < KS; ++ks)
#pragma unroll
        for (int m = 0; m < MF; ++m) w[ks][m] = a.wpk[(ks * MF + m) * 64 + lane];

    constexpr int CPW = G::NCHUNK / 4;   // DMA pieces per wave per tile
    static_assert(2 * CPW <= KS, "one DMA piece every other k-step must fit in the k-loop");
    int dma_pc[CPW];
#pragma unroll
    for (int i = 0; i < CPW; ++i) dma_pc[i] = dma_piece_const<NF>(i, wave, lane);

    __syncthreads();   // plane table / bias visible; nothing is in flight yet
    PlaneTable pt;
    pt.pl = planes_lds;
    pt.tile_begin = tile_begin_lds;
    pt.nplanes = a.nplanes;

    // bias in MFMA C/D layout (row = (reg&3) + 8*(reg>>2) + 4*half): the first MFMA of every
    // accumulator chain takes it as its C operand, so accumulators need no initialisation
    f32x16 biasv[MF];
#pragma unroll
    for (int m = 0; m < MF; ++m)
#pragma unroll
        for (int r = 0; r < 16; ++r) biasv[m][r] = bias_lds[32 * m + 8 * (r >> 2) + 4 * half + (r & 3)];

    // prologue: tile 0
    TileId id = pt.decode(t_first, lane);
    {
        const PlaneDesc& pl0 = planes_lds[id.plane];
        const char* tb = halo_tile_base<NF>(a.in_act, pl0, id.ty, id.tx);
        const int pitch0 = __builtin_amdgcn_readfirstlane(pl0.pitch) * G::PIXB;
#pragma unroll
        for (int i = 0; i < CPW; ++i) issue_dma_piece<NF>(tb, pitch0, lds0, i, wave, dma_pc[i]);
    }
    tile_barrier();

    constexpr int PF = 3;   // B fragments are read PF k-steps ahead of the MFMAs that consume them

    for (int it = 0; it < niter; ++it) {
        const int t = t_first + it * g8;
        const PlaneDesc& pl = planes_lds[id.plane];
        const char* buf = smem + (it & 1) * G::BUFB;

        // next tile: its DMA pieces are issued between this tile's MFMAs, into the other buffer
        // (free since the barrier that ended compute(it-1)).  The last iteration re-fetches its own
        // tile so that the k-loop stays one straight-line block.
        const TileId idn = pt.decode(it + 1 < niter ? t + g8 : t, lane);
        const PlaneDesc& pln = planes_lds[idn.plane];
        const char* next_tb = halo_tile_base<NF>(a.in_act, pln, idn.ty, idn.tx);
        const int next_pitch = __builtin_amdgcn_readfirstlane(pln.pitch) * G::PIXB;
        const unsigned next_lds = lds0 + ((it + 1) & 1) * G::BUFB;

        // B-operand base: pixel (row 2*wave+n, col px) of the halo tile at tap (0,0)
        const char* bbase[2];
#pragma unroll
        for (int n = 0; n < 2; ++n)
            bbase[n] = buf + ((2 * wave + n) * PW + px) * G::LPIXB + (NF == 64 ? half * 16 : 0);

        auto read_b = [&](int ks, int n) -> half8 {
            if constexpr (NF == 64) {
                // k-step ks: tap ks/4, channel octets 2*(ks%4) + half
                const int tap = ks >> 2;
                const int off = ((tap / 3) * PW + (tap % 3)) * G::LPIXB + (ks & 3) * 32;
                return *(const half8*)(bbase[n] + off);
            } else {
                // k-step ks: K octets 2ks (lanes 0-31) and 2ks+1 (lanes 32-63); octet ko is
                // tap ko/SPP, channel octet ko%SPP.  ko == KO only exists as zero weights.
                const int koA = 2 * ks, koB = (2 * ks + 1 < G::KO) ? 2 * ks + 1 : 2 * ks;
                const int tapA = koA / G::SPP, tapB = koB / G::SPP;
                const int offA = ((tapA / 3) * PW + (tapA % 3)) * G::LPIXB + (koA % G::SPP) * 16;
                const int offB = ((tapB / 3) * PW + (tapB % 3)) * G::LPIXB + (koB % G::SPP) * 16;
                return *(const half8*)(bbase[n] + (half ? offB : offA));
            }
        };

        f32x16 acc[2][MF];
        half8 bq[PF + 1][2];
        __builtin_amdgcn_sched_barrier(0);   // keep the tile decode's LDS reads out of the pipelined region
#pragma unroll
        for (int ks = 0; ks < PF; ++ks)
#pragma unroll
            for (int n = 0; n < 2; ++n) bq[ks][n] = read_b(ks, n);

#pragma unroll
        for (int ks = 0; ks < KS; ++ks) {
            if (ks + PF < KS) {
#pragma unroll
                for (int n = 0; n < 2; ++n) bq[(ks + PF) % (PF + 1)][n] = read_b(ks + PF, n);
            }
            if ((ks & 1) == 0 && ks / 2 < CPW)
                issue_dma_piece<NF>(next_tb, next_pitch, next_lds, ks / 2, wave, dma_pc[ks / 2]);
#pragma unroll
            for (int n = 0; n < 2; ++n)
#pragma unroll
                for (int m = 0; m < MF; ++m)
                    acc[n][m] = __builtin_amdgcn_mfma_f32_32x32x16_f16(w[ks][m], bq[ks % (PF + 1)][n],
                                                                        ks == 0 ? biasv[m] : acc[n][m], 0, 0, 0);
        }
        // pin the software pipeline: PF k-steps of LDS reads up front, then {MFMAs of one k-step,
        // LDS reads of one k-step}.  With one wave per SIMD nothing else hides the LDS latency.
        __builtin_amdgcn_sched_group_barrier(0x100, 2 * PF, 0);
#pragma unroll
        for (int ks = 0; ks < KS; ++ks) {
            __builtin_amdgcn_sched_group_barrier(0x008, 2 * MF, 0);
            if (ks + PF < KS) __builtin_amdgcn_sched_group_barrier(0x100, 2, 0);
        }

        // every wave is done reading buf[it&1] and this wave's share of DMA(it+1) has landed
        tile_barrier();

        if constexpr (MODE == 0) {
#pragma unroll
            for (int n = 0; n < 2; ++n)
                store_trunk<NF, MF>(acc[n], slope_lds, a.out_act, pl, id.ty * TH + 2 * wave + n,
                                    id.tx * TW + px, half);
        } else {
            const float norm = (float)(1 / 255.0);   // substract_mean_normalize norm_vals (:272, :444)
            uint8_t* stage = stage_all + wave * STAGEB;
#pragma unroll
            for (int n = 0; n < 2; ++n) {
                const int y = id.ty * TH + 2 * wave + n;
                const int x = id.tx * TW + px;
                const bool inside = (y < pl.h) && (x < pl.w);
                const int yc = min(y, pl.h - 1), xc = min(x, pl.w - 1);
#pragma unroll
                for (int m = 0; m < MF; ++m) {
#pragma unroll
                    for (int g = 0; g < 4; ++g) {
                        if (32 * m + 8 * g >= COUT) continue;
                        const int cb = 32 * m + 8 * g + 4 * half;
#pragma unroll
                        for (int j = 0; j < 4; ++j) {
                            const int co = cb + j;
                            if (co >= COUT) continue;     // per-lane (depends on half)
                            const int ch = co / (R * R);
                            const int rem = co - ch * (R * R);
                            const int i = rem / R, jj = rem - (rem / R) * R;
                            float res;
                            if constexpr (MODE == 1)
                                res = (float)a.src_u8[(size_t)(pl.src_y0 + yc) * a.src_stride +
                                                      (size_t)(pl.src_x0 + xc) * 3 + ch] * norm;
                            else
                                res = a.src_f32[((size_t)ch * pl.h + yc) * pl.w + xc];
                            const float v = acc[n][m][4 * g + j] + res;
                            if constexpr (MODE == 1) {
                                float q = __builtin_rintf(v * 255.0f);      // v_rndne_f32: ties to even
                                q = fminf(fmaxf(q, 0.f), 255.f);
                                stage[(n * R + i) * STAGE_ROWB + (px * R + jj) * 3 + ch] = (uint8_t)q;
                            } else if (inside) {
                                a.dst_f32[((size_t)ch * (pl.h * R) + (size_t)y * R + i) * ((size_t)pl.w * R) +
                                          (size_t)x * R + jj] = v;
                            }
                        }
                    }
                }
            }
            if constexpr (MODE == 1) {
                // wave-private staging -> coalesced row stores of the core region
                __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
                __builtin_amdgcn_wave_barrier();
                __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
                const int x_lo = max(pl.core_x0, id.tx * TW) - id.tx * TW;
                const int x_hi = min(min(pl.core_x1, pl.w), id.tx * TW + TW) - id.tx * TW;
                const int b_lo = x_lo * R * 3, b_hi = x_hi * R * 3;
                constexpr int WORDS = STAGE_ROWB / 4;
                for (int idx = lane; idx < 2 * R * WORDS; idx += 64) {
                    const int sr = idx / WORDS;              // staged row: n*R + i
                    const int k = idx - sr * WORDS;
                    const int n = sr / R, i = sr - n * R;
                    const int y = id.ty * TH + 2 * wave + n;
                    if (y < pl.core_y0 || y >= min(pl.core_y1, pl.h)) continue;
                    uint8_t* drow = a.dst_u8 + ((size_t)(pl.src_y0 + y) * R + i) * a.dst_stride +
                                    (size_t)(pl.src_x0 + id.tx * TW) * R * 3;
                    const uint8_t* srow = stage + sr * STAGE_ROWB;
                    const int b0 = 4 * k;
                    if (b0 >= b_lo && b0 + 4 <= b_hi && (((size_t)(drow + b0)) & 3) == 0) {
                        *(uint32_t*)(drow + b0) = *(const uint32_t*)(srow + b0);
                    } else {
#pragma unroll
                        for (int e = 0; e < 4; ++e)
                            if (b0 + e >= b_lo && b0 + e < b_hi) drow[b0 + e] = srow[b0 + e];
                    }
                }
                __builtin_amdgcn_wave_barrier();
            }
        }
        id = idn;
    }
}

// ----------------------------------------------------------------------------------------------
// head_kernel<NF, SRC>: from_pixels(PIXEL_BGR) + substract_mean_normalize + Conv_0 (3 -> NF) +
// PReLU_1, fp16 NHWC out.  SRC 0: u8 HWC source; the integer pixel values go through the MFMA
// exactly (0..255 are exact in fp16) and the 1/255 normalisation is applied to the fp32
// accumulator.  SRC 1: f32 planar source (an ncnn::Mat the caller normalised), rounded to fp16.
// K is laid out as [tap][4] (3 channels + 1 zero) -> 36, padded to 3 k-steps of 16.
// ----------------------------------------------------------------------------------------------
template <int NF, int SRC>
__global__ __launch_bounds__(256) void head_kernel(HeadArgs a)
{
    constexpr int MF = (NF + 31) / 32;
    __shared__ __attribute__((aligned(16))) char hsm[NPIX * 8 + PARAM_LDS];
    half4* tile = (half4*)hsm;
    float* bias_lds = (float*)(hsm + NPIX * 8);
    float* slope_lds = bias_lds + 64;

    const int wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
    const int lane = threadIdx.x & 63;
    const int half = lane >> 5;
    const int px = lane & 31;

    const PlaneDesc* gpl = a.planes;
    TileId id;
    id.plane = __builtin_amdgcn_readfirstlane(
        find_plane([gpl](int i) { return gpl[i].tile_begin; }, a.nplanes, (int)blockIdx.x, lane));
    const PlaneDesc& pl = a.planes[id.plane];
    {
        const int local = (int)blockIdx.x - pl.tile_begin;
        id.ty = local / pl.ntx;
        id.tx = local - id.ty * pl.ntx;
    }

    if (threadIdx.x < 64) {
        bias_lds[threadIdx.x] = threadIdx.x < MF * 32 ? a.bias[threadIdx.x] : 0.f;
        const float sl = threadIdx.x < MF * 32 ? a.slope[threadIdx.x] : 0.f;
        slope_lds[threadIdx.x] = sl;
        slope_lds[64 + threadIdx.x] = sl <= 1.f ? __builtin_inff() : -__builtin_inff();
    }
    for (int p = threadIdx.x; p < NPIX; p += 256) {
        const int r = p / PW, c = p - (p / PW) * PW;
        const int y = id.ty * TH + r - 1, x = id.tx * TW + c - 1;
        half4 v = {0, 0, 0, 0};
        if (y >= 0 && y < pl.h && x >= 0 && x < pl.w) {     // zero padding at the PLANE edge
            if constexpr (SRC == 0) {
                const uint8_t* s = a.src_u8 + (size_t)(pl.src_y0 + y) * a.src_stride + (size_t)(pl.src_x0 + x) * 3;
                v[0] = (_Float16)(float)s[0]; v[1] = (_Float16)(float)s[1]; v[2] = (_Float16)(float)s[2];
            } else {
                const size_t hw = (size_t)pl.h * pl.w, o = (size_t)y * pl.w + x;
                v[0] = (_Float16)a.src_f32[o]; v[1] = (_Float16)a.src_f32[hw + o]; v[2] = (_Float16)a.src_f32[2 * hw + o];
            }
        }
        tile[p] = v;
    }
    half8 w[3][MF];
#pragma unroll
    for (int ks = 0; ks < 3; ++ks)
#pragma unroll
        for (int m = 0; m < MF; ++m) w[ks][m] = a.wpk[(ks * MF + m) * 64 + lane];
    __syncthreads();

#pragma unroll
    for (int n = 0; n < 2; ++n) {
        f32x16 acc[MF];
#pragma unroll
        for (int m = 0; m < MF; ++m)
#pragma unroll
            for (int r = 0; r < 16; ++r) acc[m][r] = 0.f;
        const int pb = (2 * wave + n) * PW + px;
#pragma unroll
        for (int ks = 0; ks < 3; ++ks) {
            // K octet o = 2ks + half holds taps 2o and 2o+1 (tap 9.. are zero weights)
            const int o0 = 2 * ks, o1 = 2 * ks + 1;
            const int tA0 = min(2 * o0, 8), tB0 = min(2 * o0 + 1, 8);
            const int tA1 = min(2 * o1, 8), tB1 = min(2 * o1 + 1, 8);
            const int offA = half ? (tA1 / 3) * PW + tA1 % 3 : (tA0 / 3) * PW + tA0 % 3;
            const int offB = half ? (tB1 / 3) * PW + tB1 % 3 : (tB0 / 3) * PW + tB0 % 3;
            const half4 lo = tile[pb + offA], hi = tile[pb + offB];
            half8 b;
            b[0] = lo[0]; b[1] = lo[1]; b[2] = lo[2]; b[3] = lo[3];
            b[4] = hi[0]; b[5] = hi[1]; b[6] = hi[2]; b[7] = hi[3];
#pragma unroll
            for (int m = 0; m < MF; ++m)
                acc[m] = __builtin_amdgcn_mfma_f32_32x32x16_f16(w[ks][m], b, acc[m], 0, 0, 0);
        }
#pragma unroll
        for (int m = 0; m < MF; ++m)
#pragma unroll
            for (int g = 0; g < 4; ++g) {
                const f32x4 b4 = *(const f32x4*)(bias_lds + 32 * m + 8 * g + 4 * half);
#pragma unroll
                for (int j = 0; j < 4; ++j) acc[m][4 * g + j] = acc[m][4 * g + j] * a.in_scale + b4[j];
            }
        store_trunk<NF, MF>(acc, slope_lds, a.out_act, pl, id.ty * TH + 2 * wave + n, id.tx * TW + px, half);
    }
}

}  // namespace uva
