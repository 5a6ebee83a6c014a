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
// Kernels in this file (DESIGN.md section 5 has the measurements behind each choice):
//   head_kernel<NF,SRC>     u8 / f32 frame -> Conv_0 (3 -> NF) + PReLU -> fp16 NHWC planes
//   trunk_kernel<64>        one trunk layer 64 -> 64 + PReLU: 8-wave workgroup per CU, two 4-wave groups
//                           in ping-pong (k-loop of one over the epilogue of the other), weights
//                           stationary in registers, 4x32 tiles streamed through a 5-slot LDS ring by
//                           LDS-DMA, v_mfma_f32_16x16x32_f16 (the kernel is power-bound and this shape
//                           is the cheapest per flop), host-built tile schedule
//   tail_kernel<64,2>       u8 tail of the 2x net on the same skeleton (residual bytes by LDS-DMA)
//   tail4_kernel<64>        u8 tail of the 4x net: weights in LDS, 3-slot ring, stores from registers
//   conv3x3_kernel<NF,M,R>  everything else (24-feature nets, f32-route tails): one persistent 4-wave
//                           workgroup per CU (two for NF = 24), one wave per SIMD, every wave keeps the
//                           layer's whole weight matrix in its 512-entry register file, 8x32 tiles,
//                           10x34 halo tiles streamed HBM/L2 -> LDS by LDS-DMA (global_load_lds_dwordx4,
//                           no VGPR round trip) into a ring, v_mfma_f32_32x32x16_f16: A = weights (rows =
//                           output channels), B = 32 consecutive pixels of one image row, K = 16 input
//                           channels of one tap, fp32 accumulate.
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>

#include <type_traits>
#include <utility>

// In-kernel cycle stamps (s_memtime of workgroup 0 / wave 0 per phase) and the trunk kernel's ablation
// variants exist only in an instrumented build (-DUVA_INSTRUMENT: tools/trunk_anatomy.py, power_probe.py);
// the product library compiles them out.
#ifdef UVA_INSTRUMENT
#define UVA_STAMP_ON(a) ((a).dbg != nullptr && blockIdx.x == 0 && threadIdx.x == 0)
#define UVA_MEMTIME() __builtin_amdgcn_s_memtime()
#else
#define UVA_STAMP_ON(a) false
#define UVA_MEMTIME() 0ull
#endif

#include "uva_devutil.hip.h"

namespace uva {

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
    int nty4;               // work tiles of the trunk kernel (4-row tiles)
    int tile_begin4;        // first global 4-row work-tile index of this plane
};
static_assert(sizeof(PlaneDesc) == 64, "PlaneDesc layout");

template <int NF, int THT = TH>
struct Geo {
    static constexpr int NPIXT = (THT + 2) * PW;             // pixels per halo tile
    static constexpr int SPP = NF / 8;                       // 16-byte channel octets per pixel
    static constexpr int LSPP = (NF == 64) ? 9 : SPP;        // LDS slots per pixel (64ch: +1 pad slot
                                                             //  -> 144 B stride, conflict-free b128)
    static constexpr int LPIXB = LSPP * 16;                  // LDS bytes per pixel
    static constexpr int PIXB = NF * 2;                      // HBM bytes per pixel
    static constexpr int KO = 9 * SPP;                       // K octets (tap-major, then channel octet)
    static constexpr int KS = (KO + 1) / 2;                  // MFMA k-steps of 16
    static constexpr int NSLOT = NPIXT * LSPP;
    static constexpr int NCHUNK = ((NSLOT + 63) / 64 + 3) / 4 * 4;   // 1-KiB LDS-DMA pieces per halo tile,
                                                                      // a multiple of the 4 waves
    static constexpr int BUFB = NCHUNK * 1024;
};


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
    unsigned long long* dbg;      // optional: per-tile s_memtime stamps of block 0 / wave 0 (debug)
    _Float16* sink;               // >= 64 pixels of scratch: where lanes outside the image store to
    int tile_base;                // trunk_kernel: first global work tile of this launch (huge frames
                                  // are split so that a workgroup's schedule fits its LDS table)
    const uint4* sched4;          // trunk_kernel / tail_kernel: per 4-row work tile, built by the host per geometry:
                                  // x = halo origin byte offset (low 32 bits), y = offset bits 32..39 | plane << 8,
                                  // z = row pitch in bytes, w = valid columns | valid rows << 6 | tx << 9 | ty << 17
    const half8* wpk2;            // pair24_kernel: the second layer of the pair
    const float* bias2;
    const float* slope2;
    int reverse;                  // walk the tiles last-to-first: consecutive layers alternate direction so
                                  // that a layer starts on what the previous one wrote last, i.e. on what is
                                  // still in the 256 MB Infinity Cache
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
    _Float16* sink;               // headp_kernel: where lanes that own no output pixel store to (>= 1 KiB)
};

// End of a tile's k-loop: wait until all but the newest KEEP vector-memory operations of this
// wave have completed (loads, LDS-DMA included, return in issue order), drain LDS, then barrier.
template <int KEEP>
__device__ __forceinline__ void tile_barrier()
{
    static_assert(KEEP >= 0 && KEEP < 64, "vmcnt is a 6-bit field");
    asm volatile("s_waitcnt vmcnt(%0) lgkmcnt(0)\n\ts_barrier" ::"n"(KEEP) : "memory");
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
    const int* tile_begin; // LDS, [MAX_PLANES]: tile_begin (8-row tiles) or tile_begin4 (4-row tiles)
    int nplanes;
    template <bool ROWS4 = false>
    __device__ __forceinline__ TileId decode(int t, int lane) const
    {
        const int* tb = tile_begin;
        TileId id;
        id.plane = __builtin_amdgcn_readfirstlane(find_plane([tb](int i) { return tb[i]; }, nplanes, t, lane));
        const int ntx = __builtin_amdgcn_readfirstlane(pl[id.plane].ntx);
        const int local = t - __builtin_amdgcn_readfirstlane(ROWS4 ? pl[id.plane].tile_begin4 : pl[id.plane].tile_begin);
        id.ty = local / ntx;
        id.tx = local - id.ty * ntx;
        return id;
    }
};

constexpr int PLANE_LDS = MAX_PLANES * 64 + MAX_PLANES * 4;

// LDS-DMA piece i (of this wave) of the (TH+2)x(TW+2) halo tile: piece c = wave*CPW + i covers LDS
// slots [64c, 64c+64); slot q is pixel q / LSPP, octet q % LSPP.  A wave's pieces are contiguous, so
// region [wave*CPW KiB, (wave+1)*CPW KiB) of every ring slot is written by that wave only -- which
// is what lets the wave reuse the region of a consumed slot as its private output staging area.  The lane's source position is
// loop-invariant and kept packed in one register: (halo row << 16) | byte offset inside the row.
template <int NF, int THT = TH>
__device__ __forceinline__ int dma_piece_const(int i, int wave, int lane)
{
    using G = Geo<NF, THT>;
    const int q = (wave * (G::NCHUNK / 4) + i) * 64 + lane;
    int p = q / G::LSPP;
    int s = q - p * G::LSPP;
    if (s >= G::SPP) s = G::SPP - 1;          // pad slot: re-fetch the neighbouring octet
    if (p >= G::NPIXT) p = G::NPIXT - 1;      // tail of the last piece: any valid address
    const int r = p / PW;
    const int cc = p - r * PW;
    return (r << 16) | (cc * G::PIXB + s * 16);
}

template <int NF, int THT = TH>
__device__ __forceinline__ void issue_dma_piece(const char* tile_base, int pitch_bytes, unsigned lds_buf, int i,
                                                int wave, int pc)
{
    const unsigned off = (unsigned)(pc >> 16) * (unsigned)pitch_bytes + (unsigned)(pc & 0xffff);
    glds16(tile_base + off, lds_buf + (wave * (Geo<NF, THT>::NCHUNK / 4) + i) * 1024);
}

template <int NF, int THT = TH>
__device__ __forceinline__ const char* halo_tile_base(const _Float16* act, const PlaneDesc& pl, int ty, int tx)
{
    return (const char*)act +
           ((size_t)pl.act_off + (size_t)(ty * THT) * pl.pitch + (size_t)tx * TW) * Geo<NF, THT>::PIXB;
}

// Trunk epilogue shared by the head and trunk kernels: per-channel PReLU (ncnn prelu.cpp:
// x < 0 ? x*slope[c] : x), fp32 -> fp16 RNE, store into the zero-bordered NHWC plane.
// PReLU without a compare/select pair: x < 0 ? x*s : x  ==  s <= 1 ? max(x, x*s) : min(x, x*s)
// (exactly, rounding is monotonic), and both are med3(x, x*s, +-inf); prm_lds holds the slopes at
// [0,64) and the matching +-inf at [64,128).
// The MFMA C/D layout gives a lane 4 consecutive channels of ONE pixel, i.e. 8-byte pieces at a
// 128-byte stride: stored directly that is 32 cache lines per instruction and the store path, not
// HBM, becomes the limit.  So the wave's two rows (2 x 32 pixels) are transposed through a
// wave-private LDS area and leave as 16 B per lane, 1 KiB contiguous per instruction.
template <int NF>
struct StageGeo {
    static constexpr int SPX = (NF == 64) ? 144 : NF * 2;   // staging bytes per pixel (64ch: padded)
    static constexpr int BYTES = 64 * SPX;                  // per wave
};

template <int NF, int MF>
__device__ __forceinline__ void store_trunk_rows(const f32x16 (&acc)[2][MF], const float* prm_lds, char* stage,
                                                 _Float16* out_act, const PlaneDesc& pl, int y0, int x0, int lane,
                                                 unsigned long long* stamp = nullptr)
{
    constexpr int SPX = StageGeo<NF>::SPX, PIXB = NF * 2, SPP = NF / 8;
    const int px = lane & 31, half = lane >> 5;
#pragma unroll
    for (int n = 0; n < 2; ++n) {
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
                    const float x = acc[n][m][4 * g + j];
                    v[j] = __builtin_amdgcn_fmed3f(x, x * s4[j], i4[j]);
                }
                const half2v lo = __builtin_convertvector(f32x2{v[0], v[1]}, half2v);
                const half2v hi = __builtin_convertvector(f32x2{v[2], v[3]}, half2v);
                uint2 o;
                o.x = __builtin_bit_cast(unsigned, lo);
                o.y = __builtin_bit_cast(unsigned, hi);
                *(uint2*)(stage + (n * 32 + px) * SPX + cb * 2) = o;
            }
        }
    }
    __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
    __builtin_amdgcn_wave_barrier();
    __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
    if (stamp) *stamp = __builtin_amdgcn_s_memtime();
    const int vx = min(TW, pl.w - x0);          // valid pixels of this tile row (uniform)
    const int pitch = pl.pitch;
#pragma unroll
    for (int n = 0; n < 2; ++n) {
        const int y = y0 + n;
        if (y >= pl.h) continue;                // uniform
        char* grow = (char*)out_act + ((size_t)pl.act_off + (size_t)(y + 1) * pitch + (x0 + 1)) * PIXB;
#pragma unroll
        for (int k = 0; k < (32 * SPP + 63) / 64; ++k) {
            const int q = k * 64 + lane;
            const int pix = q / SPP, slot = q - pix * SPP;
            if (q < 32 * SPP && pix < vx)
                *(uint4*)(grow + pix * PIXB + slot * 16) = *(const uint4*)(stage + (n * 32 + pix) * SPX + slot * 16);
        }
    }
}

constexpr int PARAMS_AND_PLANES_LDS = 768 + MAX_PLANES * 64 + MAX_PLANES * 4;
// r = 0: trunk (no output staging); r = 1, 2, 4: tail with u8 staging of 2r rows x TW*r*3 bytes per wave
constexpr int conv_stage_bytes(int r) { return r == 0 ? 0 : 4 * (2 * r * TW * r * 3); }
template <int NF>
constexpr int conv_nbuf(int r)
{
    return 3 * Geo<NF>::BUFB + PARAMS_AND_PLANES_LDS + conv_stage_bytes(r) <= 160 * 1024 ? 3 : 2;
}
template <int NF>
constexpr int conv_lds_bytes(int r)
{
    return conv_nbuf<NF>(r) * Geo<NF>::BUFB + PARAMS_AND_PLANES_LDS + conv_stage_bytes(r);
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

    // LDS: NBUF halo-tile buffers (a ring: tile it lives in buffer it % NBUF and tile it+NBUF-1 is
    // being streamed in while tile it is computed), then parameters, plane table, tail staging.
    // trunk (24 features): 3 slots.  Tails: 2 slots -- their tile barrier then waits for ALL vector
    // memory traffic, which lets the epilogue's residual pixels be plain loads issued at the top of
    // the iteration (a deeper DMA look-ahead would force the compiler-inserted wait for those loads
    // to also wait for the youngest DMA pieces).
    constexpr int NBUF = MODE == 0 ? conv_nbuf<NF>(0) : 2;
    constexpr int LA = NBUF - 1;                       // tiles of DMA look-ahead
    extern __shared__ __attribute__((aligned(16))) char smem[];
    const unsigned lds0 = lds_offset(smem);
    float* bias_lds = (float*)(smem + NBUF * G::BUFB);
    float* slope_lds = bias_lds + 64;
    PlaneDesc* planes_lds = (PlaneDesc*)(smem + NBUF * G::BUFB + PARAM_LDS);
    int* tile_begin_lds = (int*)(smem + NBUF * G::BUFB + PARAM_LDS + MAX_PLANES * 64);
    uint8_t* stage_all = (uint8_t*)(smem + NBUF * G::BUFB + PARAM_LDS + PLANE_LDS);

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
    static_assert(CPW <= KS, "one DMA piece per k-step must fit in the k-loop");
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
    // (trunk only: the tails are short of registers and add the bias in their epilogue instead)
    f32x16 biasv[MF];
#pragma unroll
    for (int m = 0; m < MF; ++m)
#pragma unroll
        for (int r = 0; r < 16; ++r)
            biasv[m][r] = (MODE == 0) ? bias_lds[32 * m + 8 * (r >> 2) + 4 * half + (r & 3)] : 0.f;

    // prologue: tiles 0 .. LA-1 (tile indices past the end re-fetch the last tile: harmless)
    auto tile_of = [&](int i) __attribute__((always_inline)) {   // this workgroup's i-th tile
        const int t = t_first + i * g8;
        return a.reverse ? a.ntiles - 1 - t : t;
    };
    TileId ids[LA + 1];
#pragma unroll
    for (int j = 0; j < LA; ++j) {
        ids[j] = pt.decode(tile_of(min(j, niter - 1)), lane);
        const PlaneDesc& plj = planes_lds[ids[j].plane];
        const char* tb = halo_tile_base<NF>(a.in_act, plj, ids[j].ty, ids[j].tx);
        const int pitchj = __builtin_amdgcn_readfirstlane(plj.pitch) * G::PIXB;
#pragma unroll
        for (int i = 0; i < CPW; ++i) issue_dma_piece<NF>(tb, pitchj, lds0 + j * G::BUFB, i, wave, dma_pc[i]);
    }
    tile_barrier<0>();

    // B fragments are read PF k-steps ahead of the MFMAs that consume them (tails: fewer, their
    // epilogue needs the registers)
    constexpr int PF = (MODE == 0) ? 3 : (R == 4 ? 1 : 2);
    int cur = 0;            // ring slot of the tile being computed

    const bool stamp = UVA_STAMP_ON(a);
    for (int it = 0; it < niter; ++it) {
        if (stamp) a.dbg[8 * it + 0] = __builtin_amdgcn_s_memtime();
        const TileId id = ids[0];
        const PlaneDesc& pl = planes_lds[id.plane];
        const char* buf = smem + cur * G::BUFB;

        // look-ahead tile it+LA: its DMA pieces are issued between this tile's MFMAs into the ring
        // slot that compute(it-1) released at the last barrier.  Past the end the last tile is
        // re-fetched so that the k-loop stays one straight-line block.
        const int fill = cur + LA >= NBUF ? cur + LA - NBUF : cur + LA;
        ids[LA] = pt.decode(tile_of(min(it + LA, niter - 1)), lane);
        const PlaneDesc& pln = planes_lds[ids[LA].plane];
        const char* next_tb = halo_tile_base<NF>(a.in_act, pln, ids[LA].ty, ids[LA].tx);
        const int next_pitch = __builtin_amdgcn_readfirstlane(pln.pitch) * G::PIXB;
        const unsigned next_lds = lds0 + fill * G::BUFB;

        // tails: this lane's two source pixels (the residual branch of the graph), fetched now so
        // that their HBM latency (the frame was last touched ~17 layers ago) hides under the k-loop
        float resid[2][3];
        if constexpr (MODE != 0) {
#pragma unroll
            for (int n = 0; n < 2; ++n) {
                const int yc = min(id.ty * TH + 2 * wave + n, pl.h - 1), xc = min(id.tx * TW + px, pl.w - 1);
#pragma unroll
                for (int c = 0; c < 3; ++c) {
                    if constexpr (MODE == 1)
                        resid[n][c] = (float)a.src_u8[(size_t)(pl.src_y0 + yc) * a.src_stride +
                                                      (size_t)(pl.src_x0 + xc) * 3 + c];
                    else
                        resid[n][c] = a.src_f32[((size_t)c * pl.h + yc) * pl.w + xc];
                }
            }
        }

        // B-operand base: pixel (row 2*wave+n, col px) of the halo tile at tap (0,0)
        const char* bbase[2];
#pragma unroll
        for (int n = 0; n < 2; ++n)
            bbase[n] = buf + ((2 * wave + n) * PW + px) * G::LPIXB + (NF == 64 ? half * 16 : 0);

        auto read_b = [&](int ks, int n) __attribute__((always_inline)) -> half8 {
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
            if (ks < CPW) issue_dma_piece<NF>(next_tb, next_pitch, next_lds, ks, wave, dma_pc[ks]);
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

        // every wave is done reading ring slot 'cur', and this wave's share of tile it+1 has landed:
        // only the CPW pieces of each later look-ahead tile may still be in flight
        if (stamp) a.dbg[8 * it + 1] = __builtin_amdgcn_s_memtime();
        tile_barrier<(LA - 1) * CPW>();
        if (stamp) a.dbg[8 * it + 2] = __builtin_amdgcn_s_memtime();

        if constexpr (MODE == 0) {
            // staging area: this wave's own DMA region of the ring slot that was just consumed (only
            // this wave's later LDS-DMA ever writes it, and that is issued after these reads)
            static_assert(StageGeo<NF>::BYTES <= CPW * 1024, "staging must fit the wave's DMA region");
            store_trunk_rows<NF, MF>(acc, slope_lds, smem + cur * G::BUFB + wave * (CPW * 1024), a.out_act, pl,
                                     id.ty * TH + 2 * wave, id.tx * TW, opaque(lane),
                                     stamp ? a.dbg + 8 * it + 4 : nullptr);
        } else {
            const float norm = (float)(1 / 255.0);   // substract_mean_normalize norm_vals (:272, :444)
            uint8_t* stage = stage_all + wave * STAGEB;
#pragma unroll
            for (int n = 0; n < 2; ++n) {
                const int y = id.ty * TH + 2 * wave + n;
                const int x = id.tx * TW + px;
                const bool inside = (y < pl.h) && (x < pl.w);
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
                            // ch is a compile-time constant for R = 1 and 4, and depends on the
                            // lane half for R = 2 (half 0: channels 0 and 2, half 1: channel 1)
                            const float rsel = ch == 0 ? resid[n][0] : (ch == 1 ? resid[n][1] : resid[n][2]);
                            const float res = MODE == 1 ? rsel * norm : rsel;
                            const float v = (acc[n][m][4 * g + j] + bias_lds[co]) + res;
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
                if (stamp) a.dbg[8 * it + 4] = __builtin_amdgcn_s_memtime();
                const int x_lo = max(pl.core_x0, id.tx * TW) - id.tx * TW;
                const int x_hi = min(min(pl.core_x1, pl.w), id.tx * TW + TW) - id.tx * TW;
                const int b_lo = x_lo * R * 3, b_hi = x_hi * R * 3;
                constexpr int WORDS = STAGE_ROWB / 4;
                const int y_t = id.ty * TH + 2 * wave;                  // this wave's first tile row
                uint8_t* const dbase = a.dst_u8 + (size_t)(pl.src_y0 + y_t) * R * a.dst_stride +
                                       (size_t)(pl.src_x0 + id.tx * TW) * R * 3;
                // wave-uniform fast paths: both rows and all 32 columns inside the core region, and
                // every output row of the tile 16- (or 4-) byte aligned
                const bool full = b_lo == 0 && b_hi == STAGE_ROWB && y_t >= pl.core_y0 &&
                                  y_t + 2 <= min(pl.core_y1, pl.h);
                const size_t align_bits = (size_t)dbase | a.dst_stride;
                if (full && (align_bits & 15) == 0) {
                    constexpr int Q = STAGE_ROWB / 16;                  // 16-byte chunks per row
#pragma unroll
                    for (int c = 0; c < (2 * R * Q + 63) / 64; ++c) {
                        const int idx = c * 64 + lane;
                        const int sr = idx / Q, k = idx - sr * Q;
                        if (idx < 2 * R * Q)
                            *(uint4*)(dbase + (size_t)sr * a.dst_stride + 16 * k) =
                                *(const uint4*)(stage + sr * STAGE_ROWB + 16 * k);
                    }
                } else if (full && (align_bits & 3) == 0) {
#pragma unroll
                    for (int c = 0; c < (2 * R * WORDS + 63) / 64; ++c) {
                        const int idx = c * 64 + lane;
                        const int sr = idx / WORDS, k = idx - sr * WORDS;
                        if (idx < 2 * R * WORDS)
                            *(uint32_t*)(dbase + (size_t)sr * a.dst_stride + 4 * k) =
                                *(const uint32_t*)(stage + sr * STAGE_ROWB + 4 * k);
                    }
                } else {
                    for (int idx = lane; idx < 2 * R * WORDS; idx += 64) {
                        const int sr = idx / WORDS;              // staged row: n*R + i
                        const int k = idx - sr * WORDS;
                        const int n = sr / R;
                        const int y = y_t + n;
                        if (y < pl.core_y0 || y >= min(pl.core_y1, pl.h)) continue;
                        uint8_t* drow = dbase + (size_t)sr * a.dst_stride;
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
                }
                __builtin_amdgcn_wave_barrier();
            }
        }
        if (stamp) a.dbg[8 * it + 3] = __builtin_amdgcn_s_memtime();
#pragma unroll
        for (int j = 0; j < LA; ++j) ids[j] = ids[j + 1];
        cur = cur + 1 == NBUF ? 0 : cur + 1;
    }
    // LDS-DMA of the re-fetched look-ahead tiles must not outlive the workgroup's LDS allocation
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
}

// ----------------------------------------------------------------------------------------------
// pair24_kernel: TWO consecutive trunk layers of the 24-feature net (1x HurrDeblur SubCompact, 24 -> 24 -> 24,
// each + bias + PReLU) per launch.  That net is HBM-bound layer by layer (96 B of activations per pixel and layer
// against 10 kFLOP): here the intermediate image of a pair lives only in LDS, so a pair moves what one layer
// moved.  conv3x3_kernel's scheme (persistent 4-wave workgroups, two per CU, 8x32 output tiles, weights of BOTH
// layers stationary in registers, halo tiles by LDS-DMA, XCD-contiguous tile ranges), with a 2-D tile pair: the
// first layer is computed on the 10x34 halo of the output tile from a 12x36 input tile (+33 % MFMA work on that
// layer -- irrelevant for a memory-bound net, unlike for the 64-feature one, see trunk2_kernel).
//   stage A: 11 fragments of 32 pixels -- rows 0..9 x columns 0..31 of the intermediate tile plus one fragment
//            for its columns 32, 33 -- three per wave; bias + PReLU, pixels outside the plane written as ZERO (they
//            are the second layer's padding), fp16 into the LDS intermediate tile (48 B per pixel);
//   stage B: conv3x3_kernel's k-loop on that tile, output through the LDS transpose of store_trunk_rows.
// ----------------------------------------------------------------------------------------------
constexpr int P24_XH = TH + 4, P24_XW = TW + 4;                     // input halo tile of a pair: 12 x 36 pixels
constexpr int P24_PIXB = 48;                                        // 24 channels fp16, no padding (HBM and LDS)
constexpr int P24_PIECES = (P24_XH * P24_XW * 3 + 63) / 64;         // 1296 sixteen-byte slots -> 21 pieces
constexpr int P24_CPW = (P24_PIECES + 3) / 4;                       // 6 per wave (24 issued, the last 3 re-fetch)
constexpr int P24_SLOTB = 4 * P24_CPW * 1024;
constexpr int P24_INTERB = PH * PW * P24_PIXB;                      // 10 x 34 intermediate pixels
constexpr int P24_NBUF = 2;
constexpr int pair24_lds_bytes() { return P24_NBUF * P24_SLOTB + P24_INTERB + 2 * PARAM_LDS + PLANE_LDS; }
static_assert(2 * pair24_lds_bytes() <= 160 * 1024, "two pair24 workgroups per CU");
static_assert(StageGeo<24>::BYTES <= P24_CPW * 1024, "output staging must fit a wave's DMA region");

__device__ __forceinline__ int p24_piece_const(int i, int wave, int lane)
{
    const int q = (wave * P24_CPW + i) * 64 + lane;
    int p = q / 3;
    const int s = q - p * 3;
    if (p >= P24_XH * P24_XW) p = P24_XH * P24_XW - 1;      // tail of the last pieces: any valid address
    const int r = p / P24_XW, cc = p - r * P24_XW;
    return (r << 16) | (cc * P24_PIXB + s * 16);
}

__global__ __launch_bounds__(256, 2) void pair24_kernel(ConvArgs a)
{
    constexpr int NF = 24, KS = 14, SPP = 3, KO = 27, PF = 3;
    extern __shared__ __attribute__((aligned(16))) char smem[];
    const unsigned lds0 = lds_offset(smem);
    char* const inter = smem + P24_NBUF * P24_SLOTB;
    float* const prm = (float*)(inter + P24_INTERB);          // per layer: bias[64], slope[64], med3 selector[64]
    PlaneDesc* planes_lds = (PlaneDesc*)((char*)prm + 2 * PARAM_LDS);
    int* tile_begin_lds = (int*)((char*)planes_lds + MAX_PLANES * 64);

    const int wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
    const int lane = threadIdx.x & 63;
    const int half = lane >> 5;
    const int px = lane & 31;

    const int g8 = gridDim.x >> 3;
    const int xcd = blockIdx.x & 7;
    const int slot = blockIdx.x >> 3;
    const int t_first = xcd * a.tiles_per_xcd + slot;
    const int t_lim = min((xcd + 1) * a.tiles_per_xcd, a.ntiles);
    if (t_first >= t_lim) return;
    const int niter = (t_lim - t_first + g8 - 1) / g8;

    if (threadIdx.x < 64) {
        const int c = threadIdx.x;
        for (int l = 0; l < 2; ++l) {
            const float b = c < 32 ? (l ? a.bias2 : a.bias)[c] : 0.f;
            const float sl = c < 32 ? (l ? a.slope2 : a.slope)[c] : 0.f;
            prm[l * 192 + c] = b;
            prm[l * 192 + 64 + c] = sl;
            prm[l * 192 + 128 + c] = sl <= 1.f ? __builtin_inff() : -__builtin_inff();
        }
        tile_begin_lds[c] = c < a.nplanes ? a.planes[c].tile_begin : 0x7fffffff;
    }
    for (int i = threadIdx.x; i < a.nplanes * 16; i += 256) ((int*)planes_lds)[i] = ((const int*)a.planes)[i];

    // both layers' weights, resident in registers for the whole kernel (KS k-steps of 16, one 32-row block each)
    half8 wA[KS], wB[KS];
#pragma unroll
    for (int ks = 0; ks < KS; ++ks) { wA[ks] = a.wpk[ks * 64 + lane]; wB[ks] = a.wpk2[ks * 64 + lane]; }

    int dma_pc[P24_CPW];
#pragma unroll
    for (int i = 0; i < P24_CPW; ++i) dma_pc[i] = p24_piece_const(i, wave, lane);

    __syncthreads();
    PlaneTable pt;
    pt.pl = planes_lds;
    pt.tile_begin = tile_begin_lds;
    pt.nplanes = a.nplanes;

    auto tile_of = [&](int i) __attribute__((always_inline)) {
        const int t = t_first + i * g8;
        return a.reverse ? a.ntiles - 1 - t : t;
    };
    // halo origin of a pair's input tile: input pixel (ty*8 - 2, tx*32 - 2) = array position (ty*8 - 1, tx*32 - 1);
    // the activation buffers carry a guard in front of plane 0 and behind the last plane
    auto issue_tile = [&](const TileId& id, int buf) __attribute__((always_inline)) {
        const PlaneDesc& pl = planes_lds[id.plane];
        const long long pitch = __builtin_amdgcn_readfirstlane(pl.pitch);
        const char* tb = (const char*)a.in_act +
                         ((long long)pl.act_off + (long long)(id.ty * TH - 1) * pitch + (id.tx * TW - 1)) * P24_PIXB;
        tb = uniform_ptr(tb);
        const int pitchb = (int)pitch * P24_PIXB;
#pragma unroll
        for (int i = 0; i < P24_CPW; ++i) {
            const unsigned off = (unsigned)(dma_pc[i] >> 16) * (unsigned)pitchb + (unsigned)(dma_pc[i] & 0xffff);
            glds16(tb + off, lds0 + buf * P24_SLOTB + (wave * P24_CPW + i) * 1024);
        }
    };
    TileId next = pt.decode(tile_of(0), lane);
    issue_tile(next, 0);
    int cur = 0;

    // accumulator chains start from C = 0; the biases are added in the epilogues (in registers they would be 32
    // more loop-carried VGPRs next to the 112 of the two weight sets)
    f32x16 zero16;
#pragma unroll
    for (int r = 0; r < 16; ++r) zero16[r] = 0.f;

    // K octet ko = 2ks + half: tap ko / 3, channel octet ko % 3 (ko = 27 exists only as zero weights)
    auto tap_off = [&](int ks, int rowpix) __attribute__((always_inline)) -> int {
        const int koA = 2 * ks, koB = (2 * ks + 1 < KO) ? 2 * ks + 1 : 2 * ks;
        const int tA = koA / SPP, tB = koB / SPP;
        const int offA = ((tA / 3) * rowpix + (tA % 3)) * P24_PIXB + (koA % SPP) * 16;
        const int offB = ((tB / 3) * rowpix + (tB % 3)) * P24_PIXB + (koB % SPP) * 16;
        return half ? offB : offA;
    };

    const bool stamp = UVA_STAMP_ON(a);
    for (int it = 0; it < niter; ++it) {
        if (stamp) a.dbg[8 * it + 0] = __builtin_amdgcn_s_memtime();
        const TileId id = next;
        const PlaneDesc& pl = planes_lds[id.plane];
        // This wave's pieces of tile `it` have landed: they were issued BEFORE the previous tile's four output stores,
        // memory operations retire in order, so "at most 4 outstanding" proves them without waiting for the stores
        // (a store's round trip is microseconds).  After the barrier every wave's share of the tile is there and
        // nobody reads the other slot or the intermediate tile any more.
        if (it == 0) tile_barrier<0>(); else tile_barrier<4>();
        if (stamp) a.dbg[8 * it + 1] = __builtin_amdgcn_s_memtime();
        if (it + 1 < niter) {
            next = pt.decode(tile_of(it + 1), lane);
            issue_tile(next, cur ^ 1);
        }
        const char* const in_tile = smem + cur * P24_SLOTB;
        const int pl_h = __builtin_amdgcn_readfirstlane(pl.h), pl_w = __builtin_amdgcn_readfirstlane(pl.w);
        const int iy0 = id.ty * TH - 1, ix0 = id.tx * TW - 1;       // plane position of intermediate pixel (0, 0)

        // ---- stage A: intermediate fragments wave, wave + 4, wave + 8 (fragment 10 = columns 32, 33 of all rows) ----
        // per-channel parameters of this lane's 12 channels, fetched once per tile (inside the fragment loop every
        // one of these nine LDS reads is a full latency: stage A took 4 800 cycles instead of 2 000)
        f32x4 pb[3], ps[3], pi[3];
#pragma unroll
        for (int g = 0; g < 3; ++g) {
            pb[g] = *(const f32x4*)(prm + 8 * g + 4 * half);
            ps[g] = *(const f32x4*)(prm + 64 + 8 * g + 4 * half);
            pi[g] = *(const f32x4*)(prm + 128 + 8 * g + 4 * half);
        }
#pragma unroll
        for (int j = 0; j < 3; ++j) {
            const int f = wave + 4 * j;
            if (f > 10) continue;                                       // uniform
            const bool edge = f == 10;
            const int r = edge ? min(px >> 1, PH - 1) : f;
            const int c = edge ? TW + (px & 1) : px;
            const char* const base = in_tile + (r * P24_XW + c) * P24_PIXB;
            // B fragments are read PF k-steps ahead of the MFMA that consumes them (one wave per SIMD of this
            // workgroup: nothing else hides the LDS latency), order pinned with sched_group_barrier
            f32x16 acc;
            half8 bq[PF + 1];
            __builtin_amdgcn_sched_barrier(0);
#pragma unroll
            for (int ks = 0; ks < PF; ++ks) bq[ks] = *(const half8*)(base + tap_off(ks, P24_XW));
#pragma unroll
            for (int ks = 0; ks < KS; ++ks) {
                if (ks + PF < KS) bq[(ks + PF) % (PF + 1)] = *(const half8*)(base + tap_off(ks + PF, P24_XW));
                acc = __builtin_amdgcn_mfma_f32_32x32x16_f16(wA[ks], bq[ks % (PF + 1)], ks == 0 ? zero16 : acc, 0, 0, 0);
            }
            __builtin_amdgcn_sched_group_barrier(0x100, PF, 0);
#pragma unroll
            for (int ks = 0; ks < KS; ++ks) {
                __builtin_amdgcn_sched_group_barrier(0x008, 1, 0);
                if (ks + PF < KS) __builtin_amdgcn_sched_group_barrier(0x100, 1, 0);
            }
            const bool inside = (!edge || px < 2 * PH) && iy0 + r >= 0 && iy0 + r < pl_h && ix0 + c >= 0 && ix0 + c < pl_w;
            const bool write = !edge || px < 2 * PH;
            char* const dst = inter + (r * PW + c) * P24_PIXB;
#pragma unroll
            for (int g = 0; g < 3; ++g) {
                const int cb = 8 * g + 4 * half;
                const f32x4 b4 = pb[g], s4 = ps[g], i4 = pi[g];
                f32x4 v;
#pragma unroll
                for (int q = 0; q < 4; ++q) {
                    const float x = acc[4 * g + q] + b4[q];
                    v[q] = __builtin_amdgcn_fmed3f(x, x * s4[q], i4[q]);
                }
                uint2 o;
                o.x = __builtin_bit_cast(unsigned, __builtin_convertvector(f32x2{v[0], v[1]}, half2v));
                o.y = __builtin_bit_cast(unsigned, __builtin_convertvector(f32x2{v[2], v[3]}, half2v));
                if (!inside) o = make_uint2(0, 0);                      // the second layer's zero padding
                if (write) *(uint2*)(dst + cb * 2) = o;
            }
        }
        if (stamp) a.dbg[8 * it + 2] = __builtin_amdgcn_s_memtime();
        asm volatile("s_waitcnt lgkmcnt(0)\n\ts_barrier" ::: "memory");     // the intermediate tile is complete (LDS only:
                                                                              // the next tile's DMA stays in flight)
        // ---- stage B: rows 2*wave, 2*wave + 1 of the output tile from the intermediate tile ----
        if (stamp) a.dbg[8 * it + 3] = __builtin_amdgcn_s_memtime();
        f32x16 accB[2][1];
        {
            const char* const base0 = inter + ((2 * wave) * PW + px) * P24_PIXB;
            half8 bq[PF + 1][2];
            __builtin_amdgcn_sched_barrier(0);
#pragma unroll
            for (int ks = 0; ks < PF; ++ks)
#pragma unroll
                for (int n = 0; n < 2; ++n) bq[ks][n] = *(const half8*)(base0 + n * PW * P24_PIXB + tap_off(ks, PW));
#pragma unroll
            for (int ks = 0; ks < KS; ++ks) {
                if (ks + PF < KS) {
#pragma unroll
                    for (int n = 0; n < 2; ++n)
                        bq[(ks + PF) % (PF + 1)][n] = *(const half8*)(base0 + n * PW * P24_PIXB + tap_off(ks + PF, PW));
                }
#pragma unroll
                for (int n = 0; n < 2; ++n)
                    accB[n][0] = __builtin_amdgcn_mfma_f32_32x32x16_f16(wB[ks], bq[ks % (PF + 1)][n], ks == 0 ? zero16 : accB[n][0], 0, 0, 0);
            }
            __builtin_amdgcn_sched_group_barrier(0x100, 2 * PF, 0);
#pragma unroll
            for (int ks = 0; ks < KS; ++ks) {
                __builtin_amdgcn_sched_group_barrier(0x008, 2, 0);
                if (ks + PF < KS) __builtin_amdgcn_sched_group_barrier(0x100, 2, 0);
            }
        }
        if (stamp) a.dbg[8 * it + 4] = __builtin_amdgcn_s_memtime();
        f32x4 qb[3], qs[3], qi[3];
#pragma unroll
        for (int g = 0; g < 3; ++g) {
            qb[g] = *(const f32x4*)(prm + 192 + 8 * g + 4 * half);
            qs[g] = *(const f32x4*)(prm + 192 + 64 + 8 * g + 4 * half);
            qi[g] = *(const f32x4*)(prm + 192 + 128 + 8 * g + 4 * half);
        }
        // PReLU, fp16, LDS transpose (this wave's own DMA region of the slot stage A consumed: its next refill is
        // issued by this wave, after these reads), then exactly FOUR store instructions per tile whatever the tile's
        // valid extent -- lanes outside the plane store to the sink -- so that the vmcnt above is a constant
        {
            constexpr int SPX = StageGeo<NF>::SPX, PIXB = P24_PIXB;
            char* const stage = smem + cur * P24_SLOTB + wave * (P24_CPW * 1024);
            const int lane_o = opaque(lane);
            const int spx = lane_o & 31, shalf = lane_o >> 5;
#pragma unroll
            for (int n = 0; n < 2; ++n)
#pragma unroll
                for (int g = 0; g < 3; ++g) {
                    const int cb = 8 * g + 4 * shalf;
                    const f32x4 b4 = qb[g], s4 = qs[g], i4 = qi[g];
                    f32x4 v;
#pragma unroll
                    for (int q = 0; q < 4; ++q) {
                        const float x = accB[n][0][4 * g + q] + b4[q];
                        v[q] = __builtin_amdgcn_fmed3f(x, x * s4[q], i4[q]);
                    }
                    uint2 o;
                    o.x = __builtin_bit_cast(unsigned, __builtin_convertvector(f32x2{v[0], v[1]}, half2v));
                    o.y = __builtin_bit_cast(unsigned, __builtin_convertvector(f32x2{v[2], v[3]}, half2v));
                    *(uint2*)(stage + (n * 32 + spx) * SPX + cb * 2) = o;
                }
            __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
            __builtin_amdgcn_wave_barrier();
            __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
            const int y0 = id.ty * TH + 2 * wave, x0 = id.tx * TW;
            const int vx = min(TW, pl_w - x0);
            const long long pitch = __builtin_amdgcn_readfirstlane(pl.pitch);
            const long long act_off = pl.act_off;
#pragma unroll
            for (int n = 0; n < 2; ++n) {
                char* const grow = (char*)a.out_act + (act_off + (long long)(y0 + n + 1) * pitch + (x0 + 1)) * PIXB;
#pragma unroll
                for (int kk = 0; kk < 2; ++kk) {
                    const int q = min(kk * 64 + lane_o, 32 * SPP - 1);          // 96 sixteen-byte chunks per row
                    const int pix = q / SPP, sl = q - pix * SPP;
                    const bool ok = kk * 64 + lane_o < 32 * SPP && pix < vx && y0 + n < pl_h;
                    char* dst = ok ? grow + pix * PIXB + sl * 16 : (char*)a.sink + lane_o * 16;
                    *(uint4*)dst = *(const uint4*)(stage + (n * 32 + pix) * SPX + sl * 16);
                }
            }
        }
        if (stamp) a.dbg[8 * it + 5] = __builtin_amdgcn_s_memtime();
        cur ^= 1;
    }
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
}

// sub10_kernel (the whole 1x HurrDeblur net in one launch) lives in a translation unit of its own: uva_sub10.hip.h / uva_sub10.hip

// ----------------------------------------------------------------------------------------------
// trunk_kernel<NF> (NF = 64): the dominant kernel -- one trunk layer 64 -> 64 (+bias, PReLU), fp16
// NHWC in and out.
//
// One persistent 8-wave workgroup per CU = two waves per SIMD, run as two 4-wave GROUPS in
// ping-pong: while group A is in its k-loop (MFMA-bound, the matrix pipe to itself), group B runs
// the epilogue of its previous tile (PReLU, fp16 conversion, stores), waits for its LDS-DMA and
// issues nothing that needs the matrix pipe; at the next workgroup barrier the roles swap.  The
// barriers keep the two groups exactly half a period apart -- left to themselves (two independent
// workgroups per CU) they phase-lock into computing together and idling together.
//
// To fit two waves per SIMD a wave owns 32 of the 64 output channels (144 weight registers instead
// of 288) and 2 rows x 32 pixels of its group's 4-row x 32-column work tile:
//     wave-in-group = 2*rp + mh :  rows 2*rp, 2*rp+1,  output channels 32*mh .. 32*mh+31.
// A B fragment (32 pixels x 16 channels of one tap) read from LDS therefore feeds one MFMA per
// wave and is read by both mh waves: LDS read traffic is ~50 % of LDS bandwidth during a k-loop,
// the price of the register room.  Otherwise conv3x3_kernel's design: weights stationary in
// registers, halo tiles streamed by LDS-DMA into a 2-slot ring per group, zero-bordered planes.
// ----------------------------------------------------------------------------------------------
constexpr int TH4 = 4;

// LDS ring of the trunk kernel: 5 slots shared by the two groups.  Work tiles are consumed in the
// order k = 0, 1, 2, ... = g0's 1st, g1's 1st, g0's 2nd, g1's 2nd, ... (one k-loop phase each); tile
// k lives in slot k % 5.  During the k-loop of tile k the computing group streams in tile k+3 (the
// OTHER group's next-but-one k-loop) and proves it landed at the end of its own next k-loop, two
// phases later -- HBM latency under load is several microseconds here, a one-phase look-ahead left
// the DMA ~1000 cycles short.  Slot k % 5 is refilled (with tile k+5) during k-loop k+2, i.e. by the
// same group and after its epilogue of tile k, so that epilogue may use the slot as staging space.
constexpr int TRUNK_SCHED_MAX = 576;   // schedule entries (16 B each) a workgroup can hold
constexpr int TRUNK_SLOTS = 5;
constexpr int TRUNK_LOOKAHEAD = 3;
template <int NF>
struct TrunkGeo {
    using G = Geo<NF, TH4>;
    static constexpr int PIECES = (G::NSLOT + 63) / 64;      // 29 one-KiB DMA pieces per halo tile
    static constexpr int SLOTB = PIECES * 1024;
    static constexpr int CPW = (PIECES + 3) / 4;             // per wave (the last ones may repeat a piece)
    static constexpr int STAGE_PX = 80;                      // staging bytes per pixel: 32 ch fp16 + pad
    static constexpr int STAGE_WAVE = 64 * STAGE_PX;         // 2 rows x 32 pixels per wave
};
static_assert(4 * TrunkGeo<64>::STAGE_WAVE <= TrunkGeo<64>::SLOTB, "epilogue staging must fit a ring slot");
template <int NF>
constexpr int trunk_lds_bytes() { return TRUNK_SLOTS * TrunkGeo<NF>::SLOTB + PARAMS_AND_PLANES_LDS + TRUNK_SCHED_MAX * 16; }
static_assert(trunk_lds_bytes<64>() <= 160 * 1024, "trunk kernel LDS budget");

// piece i of wave 'wave': c = 4*i + wave, or a repeat of the wave's previous piece past the end
template <int NF>
__device__ __forceinline__ int trunk_piece_index(int i, int wave)
{
    const int c = 4 * i + wave;
    return c < TrunkGeo<NF>::PIECES ? c : c - 4;
}
template <int NF>
__device__ __forceinline__ int trunk_piece_const(int i, int wave, int lane)
{
    using G = Geo<NF, TH4>;
    const int q = trunk_piece_index<NF>(i, wave) * 64 + lane;
    int p = q / G::LSPP;
    int s = q - p * G::LSPP;
    if (s >= G::SPP) s = G::SPP - 1;          // pad slot: re-fetch the neighbouring octet
    if (p >= G::NPIXT) p = G::NPIXT - 1;      // tail of the last piece: any valid address
    const int r = p / PW;
    const int cc = p - r * PW;
    return (r << 16) | (cc * G::PIXB + s * 16);
}
template <int NF>
__device__ __forceinline__ void trunk_issue_piece(const char* tile_base, int pitch_bytes, unsigned lds_slot, int i,
                                                  int wave, int pc)
{
    if (4 * i + wave >= TrunkGeo<NF>::PIECES) return;     // 29 pieces, 4 waves x 8: waves 1-3 have no 8th piece
    const unsigned off = (unsigned)(pc >> 16) * (unsigned)pitch_bytes + (unsigned)(pc & 0xffff);
    glds16_s(tile_base, off, lds_slot + trunk_piece_index<NF>(i, wave) * 1024);
}
// how many of its pieces 0 .. n-1 a wave really issues
template <int NF>
__device__ __forceinline__ int trunk_pieces_issued(int n, int wave)
{
    return (n == TrunkGeo<NF>::CPW && 4 * (n - 1) + wave >= TrunkGeo<NF>::PIECES) ? n - 1 : n;
}

// ABL (debug ablations, never used by the product path): 0 = the real kernel; 1 = no MFMAs and no
// LDS reads (memory traffic only); 2 = every tile re-fetches tile 0 and stores to the sink (compute
// only, memory traffic stays in L2); 4 = memory traffic and LDS fragment reads, no MFMAs
template <int NF, int ABL = 0>
__global__ __launch_bounds__(512, 2) void trunk_kernel(ConvArgs a)
{
    static_assert(NF == 64, "the split-channel trunk kernel is written for 64 features");
    using G = Geo<NF, TH4>;
    using TG = TrunkGeo<NF>;
    constexpr int KS = G::KS;
    constexpr int CPW = TG::CPW;
    constexpr int SLOTB = TG::SLOTB;
    constexpr int PFF = 6;     // B fragments are read PFF fragments (~1.5-3 steps) ahead of their MFMAs
    constexpr int CPW_K = 5;   // DMA pieces of the look-ahead tile issued inside the k-loop (each blocks
                               // the wave's MFMA issue for ~60-150 cycles); the other CPW - CPW_K are
                               // issued by the same wave at the start of its (shorter) epilogue phase

    const unsigned long long t_entry = UVA_MEMTIME();
    extern __shared__ __attribute__((aligned(16))) char smem[];
    const unsigned lds0 = lds_offset(smem);
    float* bias_lds = (float*)(smem + TRUNK_SLOTS * SLOTB);
    float* prm_lds = bias_lds + 64;                  // slopes [0,64), med3 selectors [64,128)
    uint4* sched_lds = (uint4*)(smem + TRUNK_SLOTS * SLOTB + PARAMS_AND_PLANES_LDS);

    const int wave8 = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
    const int grp = wave8 >> 2;   // ping-pong group
    const int wave = wave8 & 3;   // wave within the group
    const int lane = threadIdx.x & 63;
    const int mh = wave & 1;      // which 32 output channels
    const int rp = wave >> 1;     // which row pair of the 4-row tile

    // persistent schedule, XCD-contiguous (see conv3x3_kernel).  Sequence number k of this
    // workgroup is work tile t0 + (k & 1) + (k >> 1) * g8: the two groups take neighbouring tiles.
    const int g8 = 2 * (gridDim.x >> 3);
    const int xcd = blockIdx.x & 7;
    const int slot = blockIdx.x >> 3;
    const int t_lim = min((xcd + 1) * a.tiles_per_xcd, a.ntiles);
    const int t0 = xcd * a.tiles_per_xcd + 2 * slot;
    if (t0 >= t_lim) return;
    const int niter0 = (t_lim - t0 + g8 - 1) / g8;
    const int niter1 = t0 + 1 < t_lim ? (t_lim - t0 - 1 + g8 - 1) / g8 : 0;
    const int niter = grp ? niter1 : niter0;

    // Tile schedule of this workgroup: its entries of the host-built per-tile table (one 16-byte
    // load per entry, requested before anything else) are copied into LDS, so the per-tile work
    // needs one uniform LDS read instead of a plane search, and none of it sits in front of a k-loop.
    const int nsched = 2 * niter0 + TRUNK_LOOKAHEAD;
    auto sched_tile = [&](int k, bool& real) __attribute__((always_inline)) {
        int t = t0 + (k & 1) + (k >> 1) * g8;
        real = t < t_lim;
        if (!real || ABL == 2) t = t0;                           // past the end: a harmless re-fetch
        if (a.reverse) t = a.ntiles - 1 - t;
        return t + a.tile_base;
    };
    uint4 sched_e0 = make_uint4(0, 0, 0, 0);
    bool sched_real0 = false;
    if ((int)threadIdx.x < nsched) sched_e0 = a.sched4[sched_tile(threadIdx.x, sched_real0)];
    float prm_b = 0.f, prm_s = 0.f;
    if (threadIdx.x < 64) { prm_b = a.bias[threadIdx.x]; prm_s = a.slope[threadIdx.x]; }
    // this wave's half of the layer's weights, resident in registers for the whole kernel.  Group 0
    // requests them right behind the schedule entries, so that they stream in under the first tiles'
    // DMA issue below.
    half8 w[KS];
    if (grp == 0) {
#pragma unroll
        for (int i = 0; i < KS; ++i) w[i] = a.wpk[((i >> 1) * 4 + 2 * mh + (i & 1)) * 64 + lane];
    }
    int dma_pc[CPW];
#pragma unroll
    for (int i = 0; i < CPW; ++i) dma_pc[i] = trunk_piece_const<NF>(i, wave, lane);
    const unsigned long long t_sched = UVA_MEMTIME();

    // accumulator chains start from C = 0 (an inline constant, no registers); the bias is added in
    // the epilogue, which has VALU slots to spare, rather than held in 16 more registers
    const f32x4 zero4 = {0.f, 0.f, 0.f, 0.f};

    char* const sink = (char*)a.sink + lane * G::PIXB;  // where lanes outside the image store to

    struct Sched { const char* base; unsigned long long off; int pitch; int vy, vx; };
    auto read_sched = [&](int k) __attribute__((always_inline)) {
        const uint4 e = sched_lds[k];
        Sched r;
        const unsigned lo = __builtin_amdgcn_readfirstlane(e.x), hi = __builtin_amdgcn_readfirstlane(e.y) & 0xffu;
        r.off = ((unsigned long long)hi << 32) | lo;
        r.base = (const char*)a.in_act + r.off;
        r.pitch = __builtin_amdgcn_readfirstlane(e.z);
        const int v = __builtin_amdgcn_readfirstlane(e.w);
        r.vy = (v >> 6) & 7; r.vx = v & 63;
        return r;
    };
    // prologue: tiles 0 and 2 by group 0, tile 1 by group 1 (all pieces).  Their schedule entries
    // come straight from the global table through the scalar cache (wave-uniform addresses): scalar
    // loads have their own counter, so the DMA issue does not queue behind the weights in flight.
    auto sched_scalar = [&](int k) __attribute__((always_inline)) {
        bool real;
        const uint4 e = scalar_load16(a.sched4 + __builtin_amdgcn_readfirstlane(sched_tile(k, real)));
        Sched r;
        const unsigned lo = __builtin_amdgcn_readfirstlane(e.x), hi = __builtin_amdgcn_readfirstlane(e.y) & 0xffu;
        r.off = ((unsigned long long)hi << 32) | lo;
        r.base = (const char*)a.in_act + r.off;
        r.pitch = __builtin_amdgcn_readfirstlane(e.z);
        r.vy = 0; r.vx = 0;
        return r;
    };
    {
        const Sched s0 = sched_scalar(grp);
#pragma unroll
        for (int i = 0; i < CPW; ++i) trunk_issue_piece<NF>(s0.base, s0.pitch, lds0 + grp * SLOTB, i, wave, dma_pc[i]);
        if (grp == 0) {
            const Sched s2 = sched_scalar(2);
#pragma unroll
            for (int i = 0; i < CPW; ++i) trunk_issue_piece<NF>(s2.base, s2.pitch, lds0 + 2 * SLOTB, i, wave, dma_pc[i]);
        }
    }
    const unsigned long long t_dma = UVA_MEMTIME();
    // everything requested so far has landed -> publish the schedule table, then the workgroup barrier
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    int real0 = sched_real0;
    asm volatile("" : "+v"(real0));     // keeps this select (and the load's wait) down here, behind the DMA issue
    if (!real0) sched_e0.w = 0;
    if ((int)threadIdx.x < nsched) sched_lds[threadIdx.x] = sched_e0;
    for (int k = threadIdx.x + 512; k < nsched; k += 512) {   // huge frames only
        bool real;
        uint4 e = a.sched4[sched_tile(k, real)];
        if (!real) e.w = 0;
        sched_lds[k] = e;
    }
    if (threadIdx.x < 64) {
        bias_lds[threadIdx.x] = prm_b;
        prm_lds[threadIdx.x] = prm_s;
        prm_lds[64 + threadIdx.x] = prm_s <= 1.f ? __builtin_inff() : -__builtin_inff();
    }
    tile_barrier<0>();
    // Group 1 starts half a period later: it fetches its weights during that wait and leaves the
    // CU's load path to group 0 until then.
    if (grp == 1) {
#pragma unroll
        for (int i = 0; i < KS; ++i) w[i] = a.wpk[((i >> 1) * 4 + 2 * mh + (i & 1)) * 64 + lane];
        group_barrier();             // group 1 runs half a period behind group 0
    }
    int cur = grp;                   // ring slot of this group's current tile: (2*it + grp) % 5
    Sched la = read_sched(grp + TRUNK_LOOKAHEAD);   // look-ahead tile of the first k-loop

    const bool stamp = UVA_STAMP_ON(a);
    if (stamp) { a.dbg[5] = t_entry; a.dbg[7] = t_sched; a.dbg[15] = t_dma; }
    for (int it = 0; it < niter0; ++it) {
        const bool active = it < niter;
        const int k = 2 * it + grp;
        f32x4 acc[2][2][2];   // [output row n][column half c][16-channel block m]
        const int fill = cur + TRUNK_LOOKAHEAD >= TRUNK_SLOTS ? cur + TRUNK_LOOKAHEAD - TRUNK_SLOTS : cur + TRUNK_LOOKAHEAD;
        const unsigned la_lds = lds0 + fill * SLOTB;   // look-ahead tile k+3 -> the slot tile k-2 left
        if (stamp) a.dbg[8 * it + 0] = __builtin_amdgcn_s_memtime();
        {
            // ---- k-loop phase: this group owns the matrix pipe; nothing but MFMAs, their LDS reads
            // and CPW_K DMA pieces is issued here ------------------------------------------------
            __builtin_amdgcn_s_setprio(2);
            const char* buf = smem + cur * SLOTB;

            // v_mfma_f32_16x16x32_f16: A = 16 output channels x 32 k (lane: channel lane&15, k octet
            // lane>>4), B = 32 k x 16 pixels (lane: pixel lane&15, k octet lane>>4), C = 4 registers.  At
            // the package power cap this shape sustains ~15 % more flop/s than 32x32x16 (half the
            // accumulator register traffic per flop; tools/mfma_power_bench.hip), and the kernel is
            // power-bound.  A k-step is (tap, input-channel half ch): a B fragment is one 16-byte LDS
            // read per lane -- pixel (halo row 2*rp + R, column 16*c + (lane&15) + dx), channels
            // 32*ch + 8*(lane>>4) .. +7.  The wave's two output rows overlap in their input rows: halo
            // row R is tap row dy = R of output row 0 and dy = R-1 of output row 1, so one fragment
            // feeds 2 (R = 0, 3) or 4 (R = 1, 2) MFMAs: 48 LDS reads per 144 MFMAs.
            //   step st = 0..23: R = st & 3, ch = (st >> 2) & 1, dx = st >> 3; fragments 2*st + c.
            // Consecutive MFMAs share their A operand (the weights) pairwise, which the power bench
            // shows to be cheaper still.
            const char* bbase = buf + ((2 * rp) * PW + (lane & 15)) * G::LPIXB + (lane >> 4) * 16;
            auto read_b = [&](int f) __attribute__((always_inline)) -> half8 {
                const int st = f >> 1, c = f & 1;
                const int R = st & 3, ch = (st >> 2) & 1, dx = st >> 3;
                return *(const half8*)(bbase + (R * PW + dx + 16 * c) * G::LPIXB + ch * 64);
            };
            constexpr int NSTEP = 24, NFRAG = 48;
            constexpr int RQ = PFF + 2;
            half8 bq[RQ];
            __builtin_amdgcn_sched_barrier(0);
#pragma unroll
            for (int f = 0; f < PFF; ++f)
                if (ABL != 1) bq[f] = read_b(f);
#pragma unroll
            for (int st = 0; st < NSTEP; ++st) {
#pragma unroll
                for (int c = 0; c < 2; ++c)
                    if (ABL != 1 && 2 * st + c + PFF < NFRAG) bq[(2 * st + c + PFF) % RQ] = read_b(2 * st + c + PFF);
                if constexpr (CPW_K > 0) {
                    constexpr int EVERY = NSTEP / (CPW_K > 0 ? CPW_K : 1);
                    if (st % EVERY == EVERY / 2 && st / EVERY < CPW_K)
                        trunk_issue_piece<NF>(la.base, la.pitch, la_lds, st / EVERY, wave, dma_pc[st / EVERY]);
                }
                const int R = st & 3, ch = (st >> 2) & 1, dx = st >> 3;
                if constexpr (ABL == 4) {      // fragments are read and consumed, but by one VALU add each
                    if (st == 0) {
#pragma unroll
                        for (int q = 0; q < 8; ++q) acc[q >> 2][(q >> 1) & 1][q & 1] = zero4;
                    }
                    const half8 b0 = bq[(2 * st) % RQ], b1 = bq[(2 * st + 1) % RQ];
                    acc[st & 1][(st >> 1) & 1][(st >> 2) & 1][st & 3] += (float)b0[st & 7] + (float)b1[(st + 3) & 7] + (float)w[st][0];
                } else if constexpr (ABL == 1) {
                    if (st == 0) {
#pragma unroll
                        for (int q = 0; q < 8; ++q) acc[q >> 2][(q >> 1) & 1][q & 1] = zero4;
                    }
                    acc[st & 1][(st >> 1) & 1][(st >> 2) & 1][st & 3] += (float)w[st][0];
                } else {
                    const half8 b0 = bq[(2 * st) % RQ], b1 = bq[(2 * st + 1) % RQ];
#pragma unroll
                    for (int n = 0; n < 2; ++n) {
                        if (n == 0 ? R > 2 : R < 1) continue;       // output row n, tap (dy = R - n, dx)
                        const bool first = st == n;                  // st 0 starts row 0, st 1 starts row 1
#pragma unroll
                        for (int m = 0; m < 2; ++m) {
                            const half8 wv = w[(((R - n) * 3 + dx) * 2 + ch) * 2 + m];
                            acc[n][0][m] = __builtin_amdgcn_mfma_f32_16x16x32_f16(wv, b0, first ? zero4 : acc[n][0][m], 0, 0, 0);
                            acc[n][1][m] = __builtin_amdgcn_mfma_f32_16x16x32_f16(wv, b1, first ? zero4 : acc[n][1][m], 0, 0, 0);
                        }
                    }
                }
            }
            // alone on its SIMD's matrix pipe, the wave must hide LDS latency itself: pin the
            // interleaving of fragment reads and MFMAs
            __builtin_amdgcn_sched_group_barrier(0x100, PFF, 0);
#pragma unroll
            for (int st = 0; st < NSTEP; ++st) {
                const int R = st & 3;
                const bool light = R == 0 || R == 3;       // 4 MFMAs; the other steps have 8
                const bool rd = 2 * st + PFF < NFRAG;
                if (light) __builtin_amdgcn_sched_group_barrier(0x008, 2, 0); else __builtin_amdgcn_sched_group_barrier(0x008, 4, 0);
                if (rd) __builtin_amdgcn_sched_group_barrier(0x100, 1, 0);
                if (light) __builtin_amdgcn_sched_group_barrier(0x008, 2, 0); else __builtin_amdgcn_sched_group_barrier(0x008, 4, 0);
                if (rd) __builtin_amdgcn_sched_group_barrier(0x100, 1, 0);
            }
            __builtin_amdgcn_s_setprio(0);
        }
        if (stamp) a.dbg[8 * it + 1] = __builtin_amdgcn_s_memtime();
        // Roles swap.  Of this wave's VMEM loads only the CPW_K pieces just issued (tile k+3) may
        // still be in flight: loads retire in order, so "at most CPW_K outstanding" proves that all
        // pieces of tile k+1 (issued one k-loop and one epilogue of this group ago, read by the other
        // group next) have landed, whatever older stores are still pending.
        tile_barrier<CPW_K>();
        if (stamp) a.dbg[8 * it + 2] = __builtin_amdgcn_s_memtime();
        // ---- epilogue phase ---------------------------------------------------------------------
        // schedule entries first (their LDS latency hides under the DMA issue below), then the rest of
        // the look-ahead tile's pieces, ahead of this phase's stores in the queue
        const Sched own = read_sched(k);
        const Sched la_next = read_sched(min(k + 2 + TRUNK_LOOKAHEAD, nsched - 1));   // for this group's next k-loop
#pragma unroll
        for (int i = CPW_K; i < CPW; ++i) trunk_issue_piece<NF>(la.base, la.pitch, la_lds, i, wave, dma_pc[i]);
        la = la_next;
        if (active) {
            // bias, PReLU (med3 form, see store_trunk_rows), fp16 RNE.  A lane holds 4 channels of one
            // pixel; stored as is, a wave-store would touch 32 cache lines with 16 bytes each and the
            // L2 request rate, not HBM, would bound the kernel.  So the wave transposes its 2 rows x
            // 32 pixels x 32 channels through the ring slot it has just finished reading and stores
            // 64 contiguous bytes per pixel.
            const int lane_o = opaque(lane);
            const int cg = lane_o >> 4, p = lane_o & 15;
            // per-channel parameters of this lane's 8 channels (4 in each 16-channel block), fetched up
            // front (one LDS latency, not eight): bias, slope, med3 selector
            f32x4 b4[2], s4[2], i4[2];
#pragma unroll
            for (int m = 0; m < 2; ++m) {
                const int cl = 32 * mh + 16 * m + 4 * cg;
                b4[m] = *(const f32x4*)(bias_lds + cl);
                s4[m] = *(const f32x4*)(prm_lds + cl);
                i4[m] = *(const f32x4*)(prm_lds + 64 + cl);
            }
            // Output pixel (row 2*rp + n, column 16*c + p) of the tile sits one row and one column inside
            // the halo origin.  After the lane exchange below, lane (p, cg) holds 8 consecutive channels
            // starting at {0, 16, 8, 24}[cg] of this wave's 32.
            char* const obase = (char*)a.out_act + own.off + ((size_t)(2 * rp + 1) * own.pitch + G::PIXB) + 64 * mh +
                                (size_t)p * G::PIXB + 32 * (cg & 1) + 8 * (cg & 2);
#pragma unroll
            for (int n = 0; n < 2; ++n)
#pragma unroll
                for (int c = 0; c < 2; ++c) {
                    unsigned o[2][2];
#pragma unroll
                    for (int m = 0; m < 2; ++m) {
                        f32x4 v;
#pragma unroll
                        for (int j = 0; j < 4; ++j) {
                            const float xv = acc[n][c][m][j] + b4[m][j];
                            v[j] = __builtin_amdgcn_fmed3f(xv, xv * s4[m][j], i4[m][j]);
                        }
                        o[m][0] = __builtin_bit_cast(unsigned, __builtin_convertvector(f32x2{v[0], v[1]}, half2v));
                        o[m][1] = __builtin_bit_cast(unsigned, __builtin_convertvector(f32x2{v[2], v[3]}, half2v));
                    }
                    // v_permlane16_swap_b32: lanes 16-31 / 48-63 of the first operand trade places with
                    // lanes 0-15 / 32-47 of the second.  Before: block m = 0 and m = 1 hold channels
                    // 4cg..4cg+3 and 16+4cg..; after: even cg lanes hold 8 consecutive channels of block 0
                    // (their own 4 and their neighbour's), odd cg lanes 8 consecutive channels of block 1.
                    const auto x = __builtin_amdgcn_permlane16_swap(o[0][0], o[1][0], false, false);
                    const auto y = __builtin_amdgcn_permlane16_swap(o[0][1], o[1][1], false, false);
                    const uint4 val = make_uint4(x[0], y[0], x[1], y[1]);
                    const bool ok = ABL != 2 && 2 * rp + n < own.vy && 16 * c + p < own.vx;
                    char* dst = ok ? obase + ((size_t)n * own.pitch + (size_t)(16 * c) * G::PIXB) : sink;
                    *(uint4*)dst = val;
                }
            if (stamp) a.dbg[8 * it + 4] = __builtin_amdgcn_s_memtime();
        }
        if (stamp) a.dbg[8 * it + 3] = __builtin_amdgcn_s_memtime();
        group_barrier();
        cur = cur + 2 >= TRUNK_SLOTS ? cur + 2 - TRUNK_SLOTS : cur + 2;
    }
    if (grp == 0) group_barrier();
    // nothing of the re-fetched look-ahead tiles may land after the workgroup's LDS is released
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    if (stamp) a.dbg[6] = __builtin_amdgcn_s_memtime();
}

// ----------------------------------------------------------------------------------------------
// trunk2_kernel<64>: TWO consecutive trunk layers (64 -> 64 -> 64, each + bias + PReLU) in one launch;
// the activations between them never leave the CU.
//
// trunk_kernel is bound by the package power cap, and a third of a launch's energy is spent on the
// 2 x 128 B per pixel that cross the Infinity Fabric.  Here the same 8-wave workgroup
// runs its two 4-wave groups as a producer / consumer pair instead of as twins:
//   group A (waves 0-3) holds the weights of layer i and computes it on 4-row x 32-column blocks of
//            the INTERMEDIATE image, written as fp16 into a 12-row ring in LDS (same 144-byte pixel
//            stride as a halo tile, so the consumer's B-fragment reads are the producer's reads);
//   group B (waves 4-7) holds the weights of layer i+1 and computes 4 rows x 30 columns of its output
//            from 6 rows x 32 columns of that ring, 2.5 steps behind A, and stores them to HBM.
// The workgroup walks DOWN a 30-column strip of a plane, so the only recomputed intermediate pixels
// are the 2 strip-edge columns (32 computed per 30 used: +6.7 % MFMA work on layer i, none on layer
// i+1) and 2 rows per strip segment; a 2-D tile pair of the same LDS footprint would recompute
// +59 % (6x34 per 4x32).  Ping-pong as in trunk_kernel: A's k-loop runs beside B's epilogue (stores),
// B's k-loop beside A's epilogue (LDS writes + ALL LDS-DMA issue for A's input tile three steps
// ahead), so neither k-loop contains anything but MFMAs and their fragment reads.
//
// Step g of a workgroup (host-built list, Trunk2Step):
//   A: intermediate rows yA .. yA+3, columns x0-1 .. x0+30 (block g % 3 of the ring) from the 6 x 34
//      input halo tile at (yA-1, x0-2) in ring slot g % 3; pixels outside the plane are written as
//      ZERO (they are layer i+1's zero padding, not layer i's output there);
//   B: output rows yA+1 .. yA+4 (= the rows whose 3x3 windows are complete once block g+1 exists),
//      columns x0 .. x0+29, from ring blocks g % 3 and (g+1) % 3; executed in iteration g+2.
// ----------------------------------------------------------------------------------------------
constexpr int T2_SW = 30;                       // output columns per strip
constexpr int T2_SLOTS = 4;                     // input halo-tile ring (A only: one tile per period; tile it+1 is complete
                                                // one barrier before its k-loop, so its first fragments are read early)
constexpr int T2_RING_ROWS = 12;                // intermediate ring: 3 blocks of 4 rows
constexpr int T2_PAD_STEPS = T2_SLOTS;          // dummy entries behind a workgroup's last step (DMA look-ahead)


struct Trunk2Args {
    const char* in_act;           // activation buffer INCLUDING its leading guard (offsets are from here)
    char* out_act;
    const half8* wpk[2];          // pack_trunk64 images of layer i and i+1
    const float* bias[2];
    const float* slope[2];
    const Trunk2Step* steps;      // [gridDim.x][max_steps + T2_PAD_STEPS]
    const int* nsteps;            // [gridDim.x]
    int max_steps;
    _Float16* sink;
    unsigned long long* dbg;      // instrumented builds: s_memtime stamps of workgroup 0, [16 * it + 8 * group + {0: k-loop start,
                                  // 1: k-loop end, 2: roles swapped, 3: epilogue end}], entry time at [16 * niter]
};

// LDS layout of trunk2_kernel: a pixel is 128 bytes = eight 16-byte slots with NO padding; slot s of the
// pixel in tile / ring column cc holds channel octet s ^ (cc & 7).  With that XOR a B-fragment read of
// v_mfma_f32_16x16x32_f16 (lane = (octet << 4) | pixel: 16 consecutive columns, 4 octets) is bank-conflict free
// for every fragment origin and both input-channel halves, which the padded 144-byte stride of trunk_kernel
// is NOT for this lane mapping: it is 2-way conflicting (profiles/r02_a_trunk_pmc.json: 47 % of the LDS
// cycles were conflict cycles).  Also 26 instead of 29 LDS-DMA pieces per input tile.
constexpr int T2_PIXB = 128;
constexpr int T2_ROWB = PW * T2_PIXB;                        // one halo-tile / ring row in LDS
constexpr int T2_PIECES = (TH4 + 2) * PW * 8 / 64 + 1;       // 6 x 34 pixels x 8 slots = 1632 slots -> 26 one-KiB pieces
constexpr int T2_SLOTB = T2_PIECES * 1024;
constexpr int T2_CPW = (T2_PIECES + 3) / 4;                  // 7 per wave (waves 2, 3: 6)
static_assert(T2_PIECES == 26 && T2_SLOTB % 128 == 0, "trunk2 tile geometry");
template <int NF>
constexpr int trunk2_lds_bytes()
{
    return T2_SLOTS * T2_SLOTB + T2_RING_ROWS * T2_ROWB + 2 * PARAM_LDS;
}
static_assert(trunk2_lds_bytes<64>() <= 160 * 1024, "trunk2 kernel LDS budget");

// LDS-DMA piece i of wave `wave`: piece c = 4i + wave covers slots [64c, 64c + 64) of the tile; slot q is
// pixel q / 8 (halo row r, column cc) and holds channel octet (q % 8) ^ (cc & 7).  -> (r << 13) | byte offset
// of that octet inside the source row.
__device__ __forceinline__ unsigned t2_piece_const(int i, int wave, int lane)
{
    const int q = (4 * i + wave) * 64 + lane;
    int pix = q >> 3;
    const int s = q & 7;
    if (pix >= (TH4 + 2) * PW) pix = (TH4 + 2) * PW - 1;      // tail of the last piece: any valid address
    const int r = pix / PW, cc = pix - r * PW;
    return (unsigned)((r << 13) | (cc * T2_PIXB + ((s ^ (cc & 7)) << 4)));
}

template <int NF>
__global__ __launch_bounds__(512, 2) void trunk2_kernel(Trunk2Args a)
{
    static_assert(NF == 64, "written for 64 features");
    using G = Geo<NF, TH4>;
    constexpr int KS = G::KS;
    constexpr int CPW = T2_CPW;
    constexpr int SLOTB = T2_SLOTB;
    constexpr int PFF = 6;
    constexpr int ROWB = T2_ROWB;
    constexpr int BLOCKB = 4 * ROWB;

    extern __shared__ __attribute__((aligned(16))) char smem[];
    const unsigned lds0 = lds_offset(smem);
    char* const ring = smem + T2_SLOTS * SLOTB;
    float* const prm_all = (float*)(ring + T2_RING_ROWS * ROWB);   // per layer: bias[64], slope[64], med3 selector[64]

    const int wave8 = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
    const int grp = wave8 >> 2;   // 0: producer (layer i), 1: consumer (layer i+1)
    const int wave = wave8 & 3;
    const int lane = threadIdx.x & 63;
    const int mh = wave & 1;      // which 32 output channels
    const int rp = wave >> 1;     // which row pair of the 4-row block

    const unsigned long long t_entry = UVA_MEMTIME();
    const int nsteps = __builtin_amdgcn_readfirstlane(a.nsteps[blockIdx.x]);
    if (nsteps <= 0) return;
    const Trunk2Step* const steps = a.steps + (size_t)blockIdx.x * (a.max_steps + T2_PAD_STEPS);
    auto load_a = [&](int g) __attribute__((always_inline)) { return scalar_load16(&steps[g].a); };
    auto load_b = [&](int g) __attribute__((always_inline)) { return scalar_load16(&steps[g].b); };

    // Prologue, ordered so that the producer's first k-loop waits for as little as possible: parameters, the
    // producer's weights and its first TWO input tiles.  The consumer's weights (not needed before iteration 2) and
    // tiles 2, 3 arrive behind the first k-loops.
    float prm_b = 0.f, prm_s = 0.f;
    if (wave == 0) {
        prm_b = (grp ? a.bias[1] : a.bias[0])[lane];
        prm_s = (grp ? a.slope[1] : a.slope[0])[lane];
    }
    // this wave's half of its layer's weights, resident in registers for the whole kernel
    half8 w[KS];
    {
        const half8* wp = grp ? a.wpk[1] : a.wpk[0];
#pragma unroll
        for (int i = 0; i < KS; ++i) w[i] = wp[((i >> 1) * 4 + 2 * mh + (i & 1)) * 64 + lane];
    }
    // LDS-DMA source position of this lane in piece i: (halo row << 13) | byte offset inside the row, two
    // pieces per register (the k-loop needs every register it can get)
    unsigned dma_pc2[(CPW + 1) / 2];
#pragma unroll
    for (int i = 0; i < (CPW + 1) / 2; ++i)
        dma_pc2[i] = t2_piece_const(2 * i, wave, lane) | (2 * i + 1 < CPW ? t2_piece_const(2 * i + 1, wave, lane) << 16 : 0u);

    auto issue_tile = [&](const uint4 e, int slot) __attribute__((always_inline)) {
        const unsigned lo = __builtin_amdgcn_readfirstlane(e.x), hi = __builtin_amdgcn_readfirstlane(e.y) & 0xffu;
        const char* base = a.in_act + (((unsigned long long)hi << 32) | lo);
        const int pitch = __builtin_amdgcn_readfirstlane(e.z);
#pragma unroll
        for (int i = 0; i < CPW; ++i) {
            if (4 * i + wave >= T2_PIECES) continue;          // 26 pieces, 4 waves x 7: waves 2, 3 have no 7th piece
            const unsigned pc = (dma_pc2[i >> 1] >> (16 * (i & 1))) & 0xffffu;
            glds16_s(base, (pc >> 13) * (unsigned)pitch + (pc & 0x1fffu), lds0 + slot * SLOTB + (4 * i + wave) * 1024);
        }
    };
    if (wave == 0) {
        float* prm = prm_all + grp * (PARAM_LDS / 4);
        prm[lane] = prm_b;
        prm[64 + lane] = prm_s;
        prm[128 + lane] = prm_s <= 1.f ? __builtin_inff() : -__builtin_inff();
    }
    if (grp == 0) {
        // input tiles 0 and 1 (entries behind the last step are valid dummies), then a "use" of the weights: hipcc
        // places its wait for them -- a vmcnt(0), which also covers the two tiles -- HERE and not in front of the
        // loop, where it would wait for tiles 2 and 3 as well
        issue_tile(load_a(0), 0);
        issue_tile(load_a(1), 1);
#pragma unroll
        for (int i = 0; i < KS; ++i) asm volatile("" : "+v"(w[i]));
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    }
    group_barrier();                   // parameters, the producer's weights and tiles 0, 1 are in place
    static_assert(T2_SLOTS == 4, "tiles 0, 1 here, tiles 2, 3 with tile 4 in the first epilogue phase");
    if (grp == 1) group_barrier();     // the consumer runs half a period behind the producer

    const float* const bias_lds = prm_all + grp * (PARAM_LDS / 4);
    const float* const prm_lds = bias_lds + 64;
    const f32x4 zero4 = {0.f, 0.f, 0.f, 0.f};

    // entries this group needs in its next epilogue phase, fetched one iteration ahead through the
    // scalar cache: A: masks of step it, input tile of step it + 3;  B: output tile of step it - 2
    uint4 e_own = grp ? make_uint4(0, 0, 0, 0) : load_a(0);
    uint4 e_dma = grp ? make_uint4(0, 0, 0, 0) : load_a(T2_SLOTS);

    // Fragment addressing of one k-loop.  A: halo tile in slot `sl`.  B: 6 ring rows starting at block bblk; the ring
    // wraps behind row 11, which only the second row pair (rows 2..5 of the window) can cross.  Lane (pixel
    // p = lane & 15, octet o = lane >> 4) reads slot (4*ch + o) ^ ((p + dx) & 7) of pixel (row 2rp + R, column
    // 16c + p + dx): the XOR term depends on dx only (16c = 0 mod 8), and the channel half flips bit 6 of an address
    // whose other terms are multiples of 128.
    struct Win { unsigned lo[3], hi[3]; };
    auto window = [&](int sl, int bblk) __attribute__((always_inline)) {
        const int pq = lane & 15, oq = lane >> 4;
        const unsigned start = grp ? (unsigned)(T2_SLOTS * SLOTB + bblk * BLOCKB) : (unsigned)(sl * SLOTB);
        const unsigned wrap = (grp && rp && bblk == 2) ? T2_RING_ROWS * ROWB : 0;
        const unsigned a0 = start + ((2 * rp) * PW + pq) * T2_PIXB;   // smem itself starts 128-byte aligned (offset 0)
        Win wn;
#pragma unroll
        for (int dx = 0; dx < 3; ++dx) {
            wn.lo[dx] = a0 + ((oq ^ ((pq + dx) & 7)) << 4);
            wn.hi[dx] = wn.lo[dx] - wrap;
        }
        return wn;
    };
    auto read_frag = [&](const Win& wn, int f) __attribute__((always_inline)) -> half8 {
        const int st = f >> 1, c = f & 1;
        const int R = st & 3, ch = (st >> 2) & 1, dx = st >> 3;
        const unsigned a = ((R < 2 ? wn.lo[dx] : wn.hi[dx]) ^ (ch ? 64u : 0u)) + (R * PW + dx + 16 * c) * T2_PIXB;
        return *(const half8*)(smem + a);
    };
    // The first PFF fragments of a k-loop are read one phase early, at the end of the group's previous epilogue phase
    // and in front of the barrier that starts the k-loop -- their LDS latency then falls into the barrier wait instead
    // of keeping the matrix pipe idle at the top of every k-loop (tools/trunk2_anatomy.py: 2 770 cycles per k-loop
    // against 144 x 17 = 2 450 of MFMA issue).  The data is there: A's tile it+1 was proven complete by ALL waves one
    // barrier earlier (4 slots: look-ahead of 4 tiles, the wait below leaves 2 in flight); B's blocks were written by
    // A two and four phases ago.
    half8 pre[PFF];
    if (grp == 0) {
        const Win w0 = window(0, 0);
#pragma unroll
        for (int f = 0; f < PFF; ++f) pre[f] = read_frag(w0, f);
    } else {
#pragma unroll
        for (int f = 0; f < PFF; ++f) pre[f] = half8{0, 0, 0, 0, 0, 0, 0, 0};
    }

    const int niter = nsteps + 2;
#ifdef UVA_INSTRUMENT
    const bool stamp = a.dbg != nullptr && blockIdx.x == 0 && (threadIdx.x & 255) == 0;
    unsigned long long* const sdbg = a.dbg + 8 * grp;
    if (stamp && grp == 0) a.dbg[16 * niter] = t_entry;
#else
    constexpr bool stamp = false;
    unsigned long long* const sdbg = nullptr;
    (void)t_entry;
#endif
    int blk = 0;                       // it % 3: A's ring block; B reads blocks blk+1, blk+2 (mod 3)
    int sl = 0;                        // it % 4: A's input slot
    for (int it = 0; it < niter; ++it) {
        const bool work = grp ? it >= 2 : it < nsteps;
        f32x4 acc[2][2][2];            // [output row n][column half c][16-channel block m]
        if (stamp) sdbg[16 * it + 0] = __builtin_amdgcn_s_memtime();
        if (work) {
            // ---- k-loop phase: MFMAs and their fragment reads, nothing else (see trunk_kernel) ----------
            __builtin_amdgcn_s_setprio(2);
            const int bblk = blk + 1 >= 3 ? blk - 2 : blk + 1;        // B: block (it - 2) % 3
            const Win wn = window(sl, bblk);
            auto read_b = [&](int f) __attribute__((always_inline)) -> half8 { return read_frag(wn, f); };
            // NARROW: the step's strip is at most 14 columns wide (the 10-column strip that ends a 970-wide plane of the
            // reference's tiling): the second fragment column (intermediate / output columns 16..31) is neither read nor
            // computed -- half of the step's MFMAs, 1.5 % of a 1080p frame's
            auto kloop = [&](auto narrow_tag) __attribute__((always_inline)) {
                constexpr bool NARROW = decltype(narrow_tag)::value;
                constexpr int NSTEP = 24, NFRAG = 48;
                constexpr int RQ = PFF + 2;
                half8 bq[RQ];
                __builtin_amdgcn_sched_barrier(0);
#pragma unroll
                for (int f = 0; f < PFF; ++f) bq[f] = pre[f];
#pragma unroll
                for (int st = 0; st < NSTEP; ++st) {
#pragma unroll
                    for (int c = 0; c < (NARROW ? 1 : 2); ++c)
                        if (2 * st + c + PFF < NFRAG) bq[(2 * st + c + PFF) % RQ] = read_b(2 * st + c + PFF);
                    const int R = st & 3, ch = (st >> 2) & 1, dx = st >> 3;
                    const half8 b0 = bq[(2 * st) % RQ], b1 = bq[(2 * st + 1) % RQ];
                    for (int n = 0; n < 2; ++n) {
                        if (n == 0 ? R > 2 : R < 1) continue;       // output row n, tap (dy = R - n, dx)
                        const bool first = st == n;
#pragma unroll
                        for (int m = 0; m < 2; ++m) {
                            const half8 wv = w[(((R - n) * 3 + dx) * 2 + ch) * 2 + m];
                            acc[n][0][m] = __builtin_amdgcn_mfma_f32_16x16x32_f16(wv, b0, first ? zero4 : acc[n][0][m], 0, 0, 0);
                            if constexpr (!NARROW)
                                acc[n][1][m] = __builtin_amdgcn_mfma_f32_16x16x32_f16(wv, b1, first ? zero4 : acc[n][1][m], 0, 0, 0);
                            else if (first) acc[n][1][m] = zero4;
                        }
                    }
                }
#pragma unroll
                for (int st = 0; st < NSTEP; ++st) {
                    const int R = st & 3;
                    const bool light = R == 0 || R == 3;
                    const bool rd = 2 * st + PFF < NFRAG;
                    constexpr int D = NARROW ? 2 : 1;
                    if (light) __builtin_amdgcn_sched_group_barrier(0x008, 2 / D, 0); else __builtin_amdgcn_sched_group_barrier(0x008, 4 / D, 0);
                    if (rd) __builtin_amdgcn_sched_group_barrier(0x100, 1, 0);
                    if (light) __builtin_amdgcn_sched_group_barrier(0x008, 2 / D, 0); else __builtin_amdgcn_sched_group_barrier(0x008, 4 / D, 0);
                    if (rd && !NARROW) __builtin_amdgcn_sched_group_barrier(0x100, 1, 0);
                }
            };
            if ((__builtin_amdgcn_readfirstlane(e_own.y) >> 25) & 1u) kloop(std::true_type{});
            else kloop(std::false_type{});
            __builtin_amdgcn_s_setprio(0);
        }
        if (stamp) sdbg[16 * it + 1] = __builtin_amdgcn_s_memtime();
        group_barrier();               // roles swap: every wave is done reading slot / blocks of this phase
        if (stamp) sdbg[16 * it + 2] = __builtin_amdgcn_s_memtime();
        // ---- epilogue phase ---------------------------------------------------------------------------
        const int lane_o = opaque(lane);
        const int cg = lane_o >> 4, p = lane_o & 15;
        // per-channel parameters of this lane's 8 channels: loaded inside the branch that uses them (defined
        // under one `if (work)` and used under another they would be loop-carried and held across the k-loop)
        auto load_params = [&](f32x4 (&b4)[2], f32x4 (&s4)[2], f32x4 (&i4)[2]) __attribute__((always_inline)) {
#pragma unroll
            for (int m = 0; m < 2; ++m) {
                const int cl = 32 * mh + 16 * m + 4 * cg;
                b4[m] = *(const f32x4*)(bias_lds + cl);
                s4[m] = *(const f32x4*)(prm_lds + cl);
                i4[m] = *(const f32x4*)(prm_lds + 64 + cl);
            }
        };
        if (grp == 0) {
            // input tile of step it + 3 -> the slot this k-loop has just released; the pieces issued one and
            // two epilogues ago (steps it + 2, it + 1) are older, so "at most two tiles' pieces outstanding"
            // at the closing barrier proves step it + 1's tile
            if (it == 0) {             // tiles 2 and 3 were left out of the prologue: hipcc's own wait for the consumer's
                issue_tile(load_a(2), 2);   // weights sits in front of the loop on BOTH groups' path and would have
                issue_tile(load_a(3), 3);   // waited for them too
            }
            issue_tile(e_dma, sl);
            if (work) {
                f32x4 b4[2], s4[2], i4[2];
                load_params(b4, s4, i4);
                const unsigned ey = __builtin_amdgcn_readfirstlane(e_own.y);
                const int rmask = (ey >> 8) & 15, c_lo = (ey >> 12) & 63, c_hi = (ey >> 18) & 63;
                // After the lane exchange (see trunk_kernel's epilogue) lane (p, cg) holds 8 consecutive channels =
                // one 16-byte slot of pixel column 16c + p: logical slot 4 mh + {0, 2, 1, 3}[cg], stored at that
                // slot ^ (p & 7).  ds_write_b128 goes out in groups of 8 consecutive lanes = 8 pixels with 8
                // different slots: conflict-free (8-byte pieces of 16 pixels were 2-way conflicting).
                char* const wbase = ring + blk * BLOCKB + ((2 * rp) * PW + p) * T2_PIXB +
                                    (((4 * mh + 2 * (cg & 1) + (cg >> 1)) ^ (p & 7)) << 4);
#pragma unroll
                for (int n = 0; n < 2; ++n)
#pragma unroll
                    for (int c = 0; c < 2; ++c) {
                        const bool inside = ((rmask >> (2 * rp + n)) & 1) && 16 * c + p >= c_lo && 16 * c + p < c_hi;
                        unsigned o[2][2];
#pragma unroll
                        for (int m = 0; m < 2; ++m) {
                            f32x4 v;
#pragma unroll
                            for (int j = 0; j < 4; ++j) {
                                const float xv = acc[n][c][m][j] + b4[m][j];
                                v[j] = __builtin_amdgcn_fmed3f(xv, xv * s4[m][j], i4[m][j]);
                            }
                            o[m][0] = __builtin_bit_cast(unsigned, __builtin_convertvector(f32x2{v[0], v[1]}, half2v));
                            o[m][1] = __builtin_bit_cast(unsigned, __builtin_convertvector(f32x2{v[2], v[3]}, half2v));
                        }
                        const auto x = __builtin_amdgcn_permlane16_swap(o[0][0], o[1][0], false, false);
                        const auto y = __builtin_amdgcn_permlane16_swap(o[0][1], o[1][1], false, false);
                        uint4 val = make_uint4(x[0], y[0], x[1], y[1]);
                        if (!inside) val = make_uint4(0, 0, 0, 0);   // layer i+1's zero padding
                        *(uint4*)(wbase + (n * PW + 16 * c) * T2_PIXB) = val;
                    }
            }
            e_own = load_a(min(it + 1, nsteps - 1));
            e_dma = load_a(it + 1 + T2_SLOTS <= nsteps + T2_PAD_STEPS - 1 ? it + 1 + T2_SLOTS : nsteps + T2_PAD_STEPS - 1);
            {   // unconditionally (a value kept across iterations would hold 24 registers through the k-loop)
                const Win wnext = window(sl + 1 == T2_SLOTS ? 0 : sl + 1, 0);
#pragma unroll
                for (int f = 0; f < PFF; ++f) pre[f] = read_frag(wnext, f);
            }
            if (stamp) sdbg[16 * it + 3] = __builtin_amdgcn_s_memtime();
            if (wave < 2) dma_barrier<2 * CPW>(); else dma_barrier<2 * (CPW - 1)>();
        } else {
            if (work) {
                f32x4 b4[2], s4[2], i4[2];
                load_params(b4, s4, i4);
                const unsigned lo = __builtin_amdgcn_readfirstlane(e_own.x), ey = __builtin_amdgcn_readfirstlane(e_own.y);
                const size_t off = ((unsigned long long)(ey & 0xffu) << 32) | lo;
                const int pitch = __builtin_amdgcn_readfirstlane(e_own.z);
                const int vy = (ey >> 8) & 7, vx = (ey >> 11) & 31;
                char* const obase = a.out_act + off + 64 * mh + (size_t)p * G::PIXB + 32 * (cg & 1) + 8 * (cg & 2) +
                                    (size_t)(2 * rp) * pitch;
#pragma unroll
                for (int n = 0; n < 2; ++n)
#pragma unroll
                    for (int c = 0; c < 2; ++c) {
                        unsigned o[2][2];
#pragma unroll
                        for (int m = 0; m < 2; ++m) {
                            f32x4 v;
#pragma unroll
                            for (int j = 0; j < 4; ++j) {
                                const float xv = acc[n][c][m][j] + b4[m][j];
                                v[j] = __builtin_amdgcn_fmed3f(xv, xv * s4[m][j], i4[m][j]);
                            }
                            o[m][0] = __builtin_bit_cast(unsigned, __builtin_convertvector(f32x2{v[0], v[1]}, half2v));
                            o[m][1] = __builtin_bit_cast(unsigned, __builtin_convertvector(f32x2{v[2], v[3]}, half2v));
                        }
                        // lane exchange: see trunk_kernel's epilogue
                        const auto x = __builtin_amdgcn_permlane16_swap(o[0][0], o[1][0], false, false);
                        const auto y = __builtin_amdgcn_permlane16_swap(o[0][1], o[1][1], false, false);
                        const uint4 val = make_uint4(x[0], y[0], x[1], y[1]);
                        const bool ok = 2 * rp + n < vy && 16 * c + p < vx;
                        char* dst = ok ? obase + ((size_t)n * pitch + (size_t)(16 * c) * G::PIXB) : (char*)a.sink + lane_o * G::PIXB;
                        *(uint4*)dst = val;
                    }
            }
            e_own = load_b(max(it - 1, 0));      // step (it + 1) - 2
            {
                const int bnext = blk + 2 >= 3 ? blk - 1 : blk + 2;       // block ((it + 1) - 2) % 3
                const Win wnext = window(0, bnext);
#pragma unroll
                for (int f = 0; f < PFF; ++f) pre[f] = read_frag(wnext, f);
            }
            if (stamp) sdbg[16 * it + 3] = __builtin_amdgcn_s_memtime();
            group_barrier();
        }
        blk = blk + 1 == 3 ? 0 : blk + 1;
        sl = sl + 1 == T2_SLOTS ? 0 : sl + 1;
    }
    if (grp == 0) group_barrier();
    // nothing of the dummy look-ahead tiles may land after the workgroup's LDS is released
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
}

// ----------------------------------------------------------------------------------------------
// tail_kernel<64, 2>: the u8 tail of the 2x net (conv 64 -> 12, PixelShuffle(2), + nearest-upsampled
// normalised input, *255, cv2 convertTo(CV_8U), core crop) on trunk_kernel's skeleton: 8-wave
// workgroup, two 4-wave groups in ping-pong, 4x32 work tiles, the same 5-slot LDS ring and tile
// schedule.  conv3x3_kernel<64,1,2> runs one wave per SIMD, so its k-loop (12 of 32 MFMA rows used)
// and its long epilogue serialise; here one group's epilogue overlaps the other's k-loop.
//   wave-in-group = 2*rp + cc: rows 2rp, 2rp+1, columns 16cc .. 16cc+15 of the tile, ALL 12 output
//   channels as one 16-row block of v_mfma_f32_16x16x32_f16 (72 weight registers).
// A lane (g = lane>>4, p = lane&15) then holds output channels 4g..4g+3 of pixel p = the 2x2
// sub-pixels of colour channel g: no cross-lane traffic in the pixel shuffle.  The residual needs
// 3 source bytes per pixel from the u8 frame: the wave fetches its 2 rows x 48 bytes as 2 x 13
// aligned dwords with one LDS-DMA instruction at the top of its k-loop phase (ordered and waited for
// like the ring pieces), instead of per-lane byte loads whose compiler-inserted wait would drain
// the DMA queue.
// ----------------------------------------------------------------------------------------------
constexpr int TAIL_SCHED_MAX = 448;
constexpr int TAIL_RESID_LDS = 8 * 128;       // per wave: 2 rows x 16 dwords
template <int NF>
constexpr int tail_lds_bytes() { return TRUNK_SLOTS * TrunkGeo<NF>::SLOTB + PARAMS_AND_PLANES_LDS + TAIL_SCHED_MAX * 16 + TAIL_RESID_LDS; }
static_assert(tail_lds_bytes<64>() <= 160 * 1024, "tail kernel LDS budget");

// 4 bytes per active lane from sbase + voff to lds_dst + lane*4
__device__ __forceinline__ void glds4_s(const void* sbase, unsigned voff, unsigned lds_dst)
{
    unsigned keep;
    asm volatile(
        "s_mov_b32 %0, m0\n\t"
        "s_mov_b32 m0, %3\n\t"
        "s_nop 0\n\t"
        "global_load_lds_dword %1, %2\n\t"
        "s_mov_b32 m0, %0"
        : "=&s"(keep)
        : "v"(voff), "s"(sbase), "s"(lds_dst));
}

template <int NF, int R>
__global__ __launch_bounds__(512, 2) void tail_kernel(ConvArgs a)
{
    static_assert(NF == 64 && R == 2, "written for the 2x net's tail (64 -> 12)");
    using G = Geo<NF, TH4>;
    using TG = TrunkGeo<NF>;
    constexpr int KS = 18;                    // k-steps of 32: (tap, input-channel half)
    constexpr int CPW = TG::CPW;
    constexpr int SLOTB = TG::SLOTB;
#define TAIL_PFF 4
    constexpr int PFF = TAIL_PFF;             // B fragments read ahead (an LDS read comes back after ~200-280 cycles, an MFMA issues in 16)
    constexpr int CPW_K = 8;                  // DMA pieces issued in the k-loop phase (the rest in the epilogue phase):
                                              // all 8 measured slightly better than 5 + 3
    constexpr int NSTEP = 24;
    constexpr int ROWB = 16 * R * 3;          // staged bytes per output row of this wave
    constexpr int STAGEB = 512;               // per wave: 2R rows x ROWB = 384
    static_assert(2 * R * ROWB <= STAGEB && 4 * STAGEB <= SLOTB, "output staging");

    extern __shared__ __attribute__((aligned(16))) char smem[];
    const unsigned lds0 = lds_offset(smem);
    float* bias_lds = (float*)(smem + TRUNK_SLOTS * SLOTB);
    PlaneDesc* planes_lds = (PlaneDesc*)(smem + TRUNK_SLOTS * SLOTB + PARAM_LDS);
    uint4* sched_lds = (uint4*)(smem + TRUNK_SLOTS * SLOTB + PARAMS_AND_PLANES_LDS);
    char* resid_all = smem + TRUNK_SLOTS * SLOTB + PARAMS_AND_PLANES_LDS + TAIL_SCHED_MAX * 16;

    const int wave8 = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
    const int grp = wave8 >> 2;
    const int wave = wave8 & 3;
    const int lane = threadIdx.x & 63;
    const int rp = wave >> 1;     // which row pair of the 4-row tile
    const int cc = wave & 1;      // which 16 columns

    const int g8 = 2 * (gridDim.x >> 3);
    const int xcd = blockIdx.x & 7;
    const int slot = blockIdx.x >> 3;
    const int t_lim = min((xcd + 1) * a.tiles_per_xcd, a.ntiles);
    const int t0 = xcd * a.tiles_per_xcd + 2 * slot;
    if (t0 >= t_lim) return;
    const int niter0 = (t_lim - t0 + g8 - 1) / g8;
    const int niter1 = t0 + 1 < t_lim ? (t_lim - t0 - 1 + g8 - 1) / g8 : 0;
    const int niter = grp ? niter1 : niter0;

    const int nsched = 2 * niter0 + TRUNK_LOOKAHEAD;
    auto sched_tile = [&](int k, bool& real) __attribute__((always_inline)) {
        int t = t0 + (k & 1) + (k >> 1) * g8;
        real = t < t_lim;
        if (!real) t = t0;                                       // past the end: a harmless re-fetch
        if (a.reverse) t = a.ntiles - 1 - t;
        return t + a.tile_base;
    };
    uint4 sched_e0 = make_uint4(0, 0, 0, 0);
    bool sched_real0 = false;
    if ((int)threadIdx.x < nsched) sched_e0 = a.sched4[sched_tile(threadIdx.x, sched_real0)];
    float prm_b = 0.f;
    if (threadIdx.x < 16) prm_b = a.bias[threadIdx.x];
    int plane_words[2] = {0, 0};
#pragma unroll
    for (int j = 0; j < 2; ++j)
        if ((int)threadIdx.x + 512 * j < a.nplanes * 16) plane_words[j] = ((const int*)a.planes)[threadIdx.x + 512 * j];
    // the layer's weights (12 output channels, zero-padded to 16), resident in registers
    half8 w[KS];
#pragma unroll
    for (int i = 0; i < KS; ++i) w[i] = a.wpk[i * 64 + lane];

    int dma_pc[CPW];
#pragma unroll
    for (int i = 0; i < CPW; ++i) dma_pc[i] = trunk_piece_const<NF>(i, wave, lane);

    const f32x4 zero4 = {0.f, 0.f, 0.f, 0.f};
    struct Sched { const char* base; int pitch; int vy, vx, ty, tx, plane; };
    auto decode = [&](const uint4 e) __attribute__((always_inline)) {
        Sched r;
        const unsigned lo = __builtin_amdgcn_readfirstlane(e.x), y = __builtin_amdgcn_readfirstlane(e.y);
        r.base = (const char*)a.in_act + (((unsigned long long)(y & 0xffu) << 32) | lo);
        r.plane = (int)(y >> 8);
        r.pitch = __builtin_amdgcn_readfirstlane(e.z);
        const unsigned v = __builtin_amdgcn_readfirstlane(e.w);
        r.vx = v & 63; r.vy = (v >> 6) & 7; r.tx = (v >> 9) & 255; r.ty = v >> 17;
        return r;
    };
    auto read_sched = [&](int k) __attribute__((always_inline)) { return decode(sched_lds[k]); };
    // prologue: tiles 0 and 2 by group 0, tile 1 by group 1; entries through the scalar cache
    {
        bool real;
        const Sched s0 = decode(scalar_load16(a.sched4 + __builtin_amdgcn_readfirstlane(sched_tile(grp, real))));
#pragma unroll
        for (int i = 0; i < CPW; ++i) trunk_issue_piece<NF>(s0.base, s0.pitch, lds0 + grp * SLOTB, i, wave, dma_pc[i]);
        if (grp == 0) {
            const Sched s2 = decode(scalar_load16(a.sched4 + __builtin_amdgcn_readfirstlane(sched_tile(2, real))));
#pragma unroll
            for (int i = 0; i < CPW; ++i) trunk_issue_piece<NF>(s2.base, s2.pitch, lds0 + 2 * SLOTB, i, wave, dma_pc[i]);
        }
    }
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    int real0 = sched_real0;
    asm volatile("" : "+v"(real0));
    if (!real0) sched_e0.w = 0;
    if ((int)threadIdx.x < nsched) sched_lds[threadIdx.x] = sched_e0;
    for (int k = threadIdx.x + 512; k < nsched; k += 512) {   // huge frames only
        bool real;
        uint4 e = a.sched4[sched_tile(k, real)];
        if (!real) e.w = 0;
        sched_lds[k] = e;
    }
    if (threadIdx.x < 16) bias_lds[threadIdx.x] = prm_b;
#pragma unroll
    for (int j = 0; j < 2; ++j)
        if ((int)threadIdx.x + 512 * j < a.nplanes * 16) ((int*)planes_lds)[threadIdx.x + 512 * j] = plane_words[j];
    tile_barrier<0>();
    if (grp == 1) group_barrier();   // group 1 runs half a period behind group 0
    int cur = grp;
    Sched la = read_sched(grp + TRUNK_LOOKAHEAD);

    const unsigned resid_lds = lds0 + (unsigned)(resid_all - smem) + wave8 * 128;
    const char* const resid_rd = resid_all + wave8 * 128;
    // the lane's four output channels' bias, resident (an LDS read at the top of every epilogue is a round trip of its own)
    const f32x4 b4 = *(const f32x4*)(bias_lds + 4 * (lane >> 4));
    const bool stamp = UVA_STAMP_ON(a);
    // A tile's schedule entry and plane fields are read (LDS, two dependent round trips) at the END of the group's previous
    // iteration, in front of the barrier it would wait at anyway: at the top of the loop they sat in front of the k-loop
    // (profiles/r04_ab_results.txt block 17).
    struct TileCtx { Sched own; int pl_h, pl_w, src_y0, src_x0, core_y0, core_y1, core_x0, core_x1; };
    auto load_ctx = [&](int k) __attribute__((always_inline)) {
        TileCtx c;
        c.own = read_sched(k);
        const PlaneDesc& pl = planes_lds[c.own.plane];
        c.pl_h = __builtin_amdgcn_readfirstlane(pl.h); c.pl_w = __builtin_amdgcn_readfirstlane(pl.w);
        c.src_y0 = __builtin_amdgcn_readfirstlane(pl.src_y0); c.src_x0 = __builtin_amdgcn_readfirstlane(pl.src_x0);
        c.core_y0 = __builtin_amdgcn_readfirstlane(pl.core_y0); c.core_y1 = __builtin_amdgcn_readfirstlane(pl.core_y1);
        c.core_x0 = __builtin_amdgcn_readfirstlane(pl.core_x0); c.core_x1 = __builtin_amdgcn_readfirstlane(pl.core_x1);
        return c;
    };
    TileCtx ctx = load_ctx(grp);
    const unsigned src_lo2 = (unsigned)(size_t)a.src_u8 & 3u, src_stride32 = (unsigned)a.src_stride;
    const unsigned dst_stride32 = (unsigned)a.dst_stride;
    for (int it = 0; it < niter0; ++it) {
        const bool active = it < niter;
        const int k = 2 * it + grp;
        if (stamp) a.dbg[8 * it + 0] = __builtin_amdgcn_s_memtime();
        const Sched own = ctx.own;
        const int pl_h = ctx.pl_h, pl_w = ctx.pl_w, src_y0 = ctx.src_y0, src_x0 = ctx.src_x0;
        const int y_t = own.ty * TH4 + 2 * rp;            // plane-local first row of this wave
        const int xs = own.tx * TW + 16 * cc;             // plane-local first column of this wave
        // residual source bytes: rows y_t, y_t+1 (clamped), columns xs .. xs+15 (clamped) of the plane.  Offsets inside the
        // frame are 32-bit (the DMA's lane offset is): uva_net_process_u8 refuses frames of 4 GB and more
        int sh[2];
        {
            const int xc = min(xs, pl_w - 1);
            const unsigned nb = 3u * (unsigned)min(16, pl_w - xc);         // valid bytes of the row segment (>= 3)
            unsigned voff = 0;
#pragma unroll
            for (int n = 0; n < 2; ++n) {
                const int yc = min(y_t + n, pl_h - 1);
                const unsigned ra = (unsigned)(src_y0 + yc) * src_stride32 + (unsigned)(src_x0 + xc) * 3u;   // byte offset in the frame
                const unsigned al = src_lo2 + ra;                          // same low bits as the byte's address
                sh[n] = (int)(al & 3u);
                const int dmax = (int)((((al + nb - 1u) & ~3u) - (al & ~3u)) >> 2);
                const unsigned vo = (ra - (unsigned)sh[n]) + 4u * (unsigned)min(lane & 15, dmax);
                if ((lane >> 4) == n) voff = vo;
            }
            if (active && lane < 32) glds4_s(a.src_u8, voff, resid_lds);
        }
        f32x4 acc[2];
        const int fill = cur + TRUNK_LOOKAHEAD >= TRUNK_SLOTS ? cur + TRUNK_LOOKAHEAD - TRUNK_SLOTS : cur + TRUNK_LOOKAHEAD;
        const unsigned la_lds = lds0 + fill * SLOTB;
        {
            // ---- k-loop phase: 24 fragment reads, 36 MFMAs, CPW_K DMA pieces of the look-ahead tile ----
            __builtin_amdgcn_s_setprio(2);
            const char* buf = smem + cur * SLOTB;
            const char* bbase = buf + ((2 * rp) * PW + 16 * cc + (lane & 15)) * G::LPIXB + (lane >> 4) * 16;
            auto read_b = [&](int st) __attribute__((always_inline)) -> half8 {
                const int Rr = st & 3, ch = (st >> 2) & 1, dx = st >> 3;
                return *(const half8*)(bbase + (Rr * PW + dx) * G::LPIXB + ch * 64);
            };
            constexpr int RQ = PFF + 1;
            half8 bq[RQ];
            __builtin_amdgcn_sched_barrier(0);
#pragma unroll
            for (int f = 0; f < PFF; ++f) bq[f] = read_b(f);
            __builtin_amdgcn_sched_barrier(0);
            // The order is pinned step by step (sched_barrier): the read PFF fragments ahead, a DMA piece, this fragment's
            // MFMAs.  Left to itself hipcc sinks every read to just in front of its MFMA ("read, wait, MFMA": sixteen
            // lgkmcnt(0) waits per k-loop, each an LDS round trip of ~200 cycles -- the k-loop then takes 2 190 ticks for
            // 576 cycles of MFMA issue, profiles/r04_ab_results.txt block 17).
            static_for<NSTEP>([&](auto sc) __attribute__((always_inline)) {
                constexpr int st = decltype(sc)::value;
                if constexpr (st + PFF < NSTEP) bq[(st + PFF) % RQ] = read_b(st + PFF);
                constexpr int EVERY = NSTEP / CPW_K;
                if constexpr (st % EVERY == 1 && st / EVERY < CPW_K)
                    trunk_issue_piece<NF>(la.base, la.pitch, la_lds, st / EVERY, wave, dma_pc[st / EVERY]);
                constexpr int Rr = st & 3, ch = (st >> 2) & 1, dx = st >> 3;
                const half8 b = bq[st % RQ];
                if constexpr (Rr <= 2)    // output row 0, tap (dy = Rr, dx)
                    acc[0] = __builtin_amdgcn_mfma_f32_16x16x32_f16(w[((Rr * 3 + dx) * 2) + ch], b, st == 0 ? zero4 : acc[0], 0, 0, 0);
                if constexpr (Rr >= 1)    // output row 1, tap (dy = Rr - 1, dx)
                    acc[1] = __builtin_amdgcn_mfma_f32_16x16x32_f16(w[(((Rr - 1) * 3 + dx) * 2) + ch], b, st == 1 ? zero4 : acc[1], 0, 0, 0);
                __builtin_amdgcn_sched_barrier(0);
            });
            __builtin_amdgcn_s_setprio(0);
        }
        if (stamp) a.dbg[8 * it + 1] = __builtin_amdgcn_s_memtime();
        // only the pieces of tile k+3 just issued (CPW_K, one fewer for waves without an 8th piece) may
        // still be in flight: the residual dwords (issued before them) and tile k+1 have landed
        if (trunk_pieces_issued<NF>(CPW_K, wave) == CPW_K) tile_barrier<CPW_K>();
        else tile_barrier<CPW_K - 1>();
        if (stamp) a.dbg[8 * it + 2] = __builtin_amdgcn_s_memtime();
        // ---- epilogue phase ----------------------------------------------------------------------
#pragma unroll
        for (int i = CPW_K; i < CPW; ++i) trunk_issue_piece<NF>(la.base, la.pitch, la_lds, i, wave, dma_pc[i]);
        if (active) {
            const int lane_o = lane;
            const int g = lane_o >> 4, p = lane_o & 15;          // colour channel (3: padding rows), pixel
            uint8_t* const stage = (uint8_t*)(smem + cur * SLOTB + wave * STAGEB);
            const float norm = (float)(1 / 255.0);               // substract_mean_normalize norm_vals (:272, :444)
            if (g < 3) {
                unsigned rb[2];
#pragma unroll
                for (int n = 0; n < 2; ++n) rb[n] = *(const uint8_t*)(resid_rd + n * 64 + sh[n] + 3 * p + g);
#pragma unroll
                for (int n = 0; n < 2; ++n) {
                    const float res = (float)rb[n] * norm;
                    // v_cvt_pk_u8_f32 rounds half to even and saturates: cv2's convertTo(CV_8U) in one instruction (as in sub10_kernel)
                    unsigned q4 = 0;
#pragma unroll
                    for (int j = 0; j < 4; ++j) q4 = __builtin_amdgcn_cvt_pk_u8_f32(((acc[n][j] + b4[j]) + res) * 255.0f, j, q4);
#pragma unroll
                    for (int j = 0; j < 4; ++j)
                        stage[(n * R + (j >> 1)) * ROWB + (p * R + (j & 1)) * 3 + g] = (uint8_t)(q4 >> (8 * j));
                }
            }
            __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
            __builtin_amdgcn_wave_barrier();
            __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
            if (stamp) a.dbg[8 * it + 4] = __builtin_amdgcn_s_memtime();
            // copy-out of the plane's core region (process_tile's crop, upscale_processing.py:464-477)
            const int core_y0 = ctx.core_y0, core_y1 = min(ctx.core_y1, pl_h);
            const int core_x0 = ctx.core_x0, core_x1 = min(ctx.core_x1, pl_w);
            const int x_lo = max(core_x0, xs) - xs, x_hi = min(core_x1, xs + 16) - xs;
            const int b_lo = x_lo * R * 3, b_hi = x_hi * R * 3;
            // (32-bit offset inside the output frame: uva_net_process_u8_device refuses frames of 4 GB and more)
            uint8_t* const dbase = a.dst_u8 + ((unsigned)(src_y0 + y_t) * (unsigned)R * dst_stride32 + (unsigned)(src_x0 + xs) * (unsigned)(R * 3));
            const bool full = b_lo == 0 && b_hi == ROWB && y_t >= core_y0 && y_t + 2 <= core_y1;
            const size_t align_bits = (size_t)dbase | a.dst_stride;
            constexpr int Q = ROWB / 16, WORDS = ROWB / 4;
            if (full && (align_bits & 15) == 0) {
                if (lane_o < 2 * R * Q) {
                    const int sr = lane_o / Q, kq = lane_o - sr * Q;
                    *(uint4*)(dbase + (unsigned)sr * dst_stride32 + 16 * kq) = *(const uint4*)(stage + sr * ROWB + 16 * kq);
                }
            } else if (b_hi > b_lo) {
                for (int idx = lane_o; idx < 2 * R * WORDS; idx += 64) {
                    const int sr = idx / WORDS, kw = idx - sr * WORDS;
                    const int y = y_t + sr / R;
                    if (y < core_y0 || y >= core_y1) continue;
                    uint8_t* drow = dbase + (size_t)sr * a.dst_stride;
                    const uint8_t* srow = stage + sr * ROWB;
                    const int b0 = 4 * kw;
                    if (b0 >= b_lo && b0 + 4 <= b_hi && (((size_t)(drow + b0)) & 3) == 0) {
                        *(uint32_t*)(drow + b0) = *(const uint32_t*)(srow + b0);
                    } else {
#pragma unroll
                        for (int e = 0; e < 4; ++e)
                            if (b0 + e >= b_lo && b0 + e < b_hi) drow[b0 + e] = srow[b0 + e];
                    }
                }
            }
            __builtin_amdgcn_wave_barrier();
        }
        if (stamp) a.dbg[8 * it + 3] = __builtin_amdgcn_s_memtime();
        ctx = load_ctx(min(k + 2, nsched - 1));           // the group's next tile (past the end: an inert entry)
        la = read_sched(min(k + 2 + TRUNK_LOOKAHEAD, nsched - 1));      // ... and the tile its k-loop will fetch
        group_barrier();
        cur = cur + 2 >= TRUNK_SLOTS ? cur + 2 - TRUNK_SLOTS : cur + 2;
    }
    if (grp == 0) group_barrier();
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
}

// ----------------------------------------------------------------------------------------------
// tail4_kernel<64>: the u8 tail of the 4x net (conv 64 -> 48, PixelShuffle(4), + nearest-upsampled
// normalised input, *255, cv2 convertTo(CV_8U), core crop), again two 4-wave groups in ping-pong on
// 4x32 tiles.  48 output channels = three 16-row blocks, one per colour channel: with the 2x tail's
// pixel split (wave = 2 rows x 16 columns) the weights would be 216 registers per wave, too many for
// two waves per SIMD, so they live in LDS (54 KB, an A fragment is one linear ds_read_b128 per lane)
// and the ring shrinks to 3 slots:
//   * tile k sits in slot k % 3; the group that has just finished k-loop k refills that very slot
//     with tile k+3 (the other group's next-but-one) at the END of its epilogue phase, after its
//     output stores, and issues nothing but the residual dwords during its k-loop -- so "everything
//     has landed" (vmcnt(0)) at the end of the next k-loop is a cheap wait and proves tile k+3 in
//     time (1.x phases of latency hiding);
//   * a lane (g, p) holds, over the three blocks, sub-row g and sub-columns 0..3 of all three colour
//     channels of pixel p: 12 contiguous output bytes, stored straight from registers.
// ----------------------------------------------------------------------------------------------
constexpr int TAIL4_SLOTS = 3;
constexpr int TAIL4_MB = 3;                                  // 16-channel blocks = colour channels
constexpr int TAIL4_W_LDS = 18 * TAIL4_MB * 1024;            // weights: [k-step][block][lane][8] fp16
template <int NF>
constexpr int tail4_lds_bytes()
{
    return TAIL4_SLOTS * TrunkGeo<NF>::SLOTB + TAIL4_W_LDS + PARAMS_AND_PLANES_LDS + TAIL_SCHED_MAX * 16 + TAIL_RESID_LDS;
}
static_assert(tail4_lds_bytes<64>() <= 160 * 1024, "tail4 kernel LDS budget");

template <int NF>
__global__ __launch_bounds__(512, 2) void tail4_kernel(ConvArgs a)
{
    static_assert(NF == 64, "written for the 4x net's tail (64 -> 48)");
    using G = Geo<NF, TH4>;
    using TG = TrunkGeo<NF>;
    constexpr int R = 4, MB = TAIL4_MB;
    constexpr int CPW = TG::CPW;
    constexpr int SLOTB = TG::SLOTB;
#define TAIL4_PFF 6
#define TAIL4_AD 2
    constexpr int PFF = TAIL4_PFF;            // B fragments read ahead
    constexpr int NSTEP = 24;

    extern __shared__ __attribute__((aligned(16))) char smem[];
    const unsigned lds0 = lds_offset(smem);
    char* w_lds = smem + TAIL4_SLOTS * SLOTB;
    char* const after_w = w_lds + TAIL4_W_LDS;
    float* bias_lds = (float*)after_w;
    PlaneDesc* planes_lds = (PlaneDesc*)(after_w + PARAM_LDS);
    uint4* sched_lds = (uint4*)(after_w + PARAMS_AND_PLANES_LDS);
    char* resid_all = after_w + PARAMS_AND_PLANES_LDS + TAIL_SCHED_MAX * 16;

    const int wave8 = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
    const int grp = wave8 >> 2;
    const int wave = wave8 & 3;
    const int lane = threadIdx.x & 63;
    const int rp = wave >> 1;     // which row pair of the 4-row tile
    const int cc = wave & 1;      // which 16 columns

    const int g8 = 2 * (gridDim.x >> 3);
    const int xcd = blockIdx.x & 7;
    const int slot = blockIdx.x >> 3;
    const int t_lim = min((xcd + 1) * a.tiles_per_xcd, a.ntiles);
    const int t0 = xcd * a.tiles_per_xcd + 2 * slot;
    if (t0 >= t_lim) return;
    const int niter0 = (t_lim - t0 + g8 - 1) / g8;
    const int niter1 = t0 + 1 < t_lim ? (t_lim - t0 - 1 + g8 - 1) / g8 : 0;
    const int niter = grp ? niter1 : niter0;

    const int nsched = 2 * niter0 + TRUNK_LOOKAHEAD;
    auto sched_tile = [&](int k, bool& real) __attribute__((always_inline)) {
        int t = t0 + (k & 1) + (k >> 1) * g8;
        real = t < t_lim;
        if (!real) t = t0;                                       // past the end: a harmless re-fetch
        if (a.reverse) t = a.ntiles - 1 - t;
        return t + a.tile_base;
    };
    int dma_pc[CPW];
#pragma unroll
    for (int i = 0; i < CPW; ++i) dma_pc[i] = trunk_piece_const<NF>(i, wave, lane);

    struct Sched { const char* base; int pitch; int vy, vx, ty, tx, plane; };
    auto decode = [&](const uint4 e) __attribute__((always_inline)) {
        Sched r;
        const unsigned lo = __builtin_amdgcn_readfirstlane(e.x), y = __builtin_amdgcn_readfirstlane(e.y);
        r.base = (const char*)a.in_act + (((unsigned long long)(y & 0xffu) << 32) | lo);
        r.plane = (int)(y >> 8);
        r.pitch = __builtin_amdgcn_readfirstlane(e.z);
        const unsigned v = __builtin_amdgcn_readfirstlane(e.w);
        r.vx = v & 63; r.vy = (v >> 6) & 7; r.tx = (v >> 9) & 255; r.ty = v >> 17;
        return r;
    };
    auto read_sched = [&](int k) __attribute__((always_inline)) { return decode(sched_lds[k]); };
    // prologue: the first three tiles' DMA (entries through the scalar cache), then tables and weights
    {
        bool real;
        const Sched s0 = decode(scalar_load16(a.sched4 + __builtin_amdgcn_readfirstlane(sched_tile(grp, real))));
#pragma unroll
        for (int i = 0; i < CPW; ++i) trunk_issue_piece<NF>(s0.base, s0.pitch, lds0 + grp * SLOTB, i, wave, dma_pc[i]);
        if (grp == 0) {
            const Sched s2 = decode(scalar_load16(a.sched4 + __builtin_amdgcn_readfirstlane(sched_tile(2, real))));
#pragma unroll
            for (int i = 0; i < CPW; ++i) trunk_issue_piece<NF>(s2.base, s2.pitch, lds0 + 2 * SLOTB, i, wave, dma_pc[i]);
        }
    }
    for (int k = threadIdx.x; k < nsched; k += 512) {
        bool real;
        uint4 e = a.sched4[sched_tile(k, real)];
        if (!real) e.w = 0;
        sched_lds[k] = e;
    }
    if (threadIdx.x < 48) bias_lds[threadIdx.x] = a.bias[threadIdx.x];
    for (int i = threadIdx.x; i < a.nplanes * 16; i += 512) ((int*)planes_lds)[i] = ((const int*)a.planes)[i];
    for (int i = threadIdx.x; i < TAIL4_W_LDS / 16; i += 512) ((uint4*)w_lds)[i] = ((const uint4*)a.wpk)[i];
    tile_barrier<0>();
    if (grp == 1) group_barrier();   // group 1 runs half a period behind group 0
    int cur = grp;                   // ring slot of this group's current tile: (2*it + grp) % 3

    const unsigned resid_lds = lds0 + (unsigned)(resid_all - smem) + wave8 * 128;
    const char* const resid_rd = resid_all + wave8 * 128;
    const char* const wrd = w_lds + lane * 16;
    const f32x4 zero4 = {0.f, 0.f, 0.f, 0.f};
    // the lane's 3 x 4 output channels' bias, resident (no LDS round trip at the top of every epilogue)
    f32x4 b4[MB];
#pragma unroll
    for (int m = 0; m < MB; ++m) b4[m] = *(const f32x4*)(bias_lds + 16 * m + 4 * (lane >> 4));
    // the next tile's schedule entry and plane fields are read at the end of the previous iteration (see tail_kernel)
    struct TileCtx { Sched own; int pl_h, pl_w, src_y0, src_x0, core_y0, core_y1, core_x0, core_x1; };
    auto load_ctx = [&](int k) __attribute__((always_inline)) {
        TileCtx c;
        c.own = read_sched(k);
        const PlaneDesc& pl = planes_lds[c.own.plane];
        c.pl_h = __builtin_amdgcn_readfirstlane(pl.h); c.pl_w = __builtin_amdgcn_readfirstlane(pl.w);
        c.src_y0 = __builtin_amdgcn_readfirstlane(pl.src_y0); c.src_x0 = __builtin_amdgcn_readfirstlane(pl.src_x0);
        c.core_y0 = __builtin_amdgcn_readfirstlane(pl.core_y0); c.core_y1 = __builtin_amdgcn_readfirstlane(pl.core_y1);
        c.core_x0 = __builtin_amdgcn_readfirstlane(pl.core_x0); c.core_x1 = __builtin_amdgcn_readfirstlane(pl.core_x1);
        return c;
    };
    TileCtx ctx = load_ctx(grp);
    const unsigned src_lo2 = (unsigned)(size_t)a.src_u8 & 3u, src_stride32 = (unsigned)a.src_stride;
    for (int it = 0; it < niter0; ++it) {
        const bool active = it < niter;
        const int k = 2 * it + grp;
        const Sched own = ctx.own;
        const int pl_h = ctx.pl_h, pl_w = ctx.pl_w, src_y0 = ctx.src_y0, src_x0 = ctx.src_x0;
        const int y_t = own.ty * TH4 + 2 * rp;            // plane-local first row of this wave
        const int xs = own.tx * TW + 16 * cc;             // plane-local first column of this wave
        // residual source bytes: rows y_t, y_t+1 (clamped), columns xs .. xs+15 (clamped), 2 x 13 dwords; 32-bit frame offsets
        int sh[2];
        {
            const int xc = min(xs, pl_w - 1);
            const unsigned nb = 3u * (unsigned)min(16, pl_w - xc);
            unsigned voff = 0;
#pragma unroll
            for (int n = 0; n < 2; ++n) {
                const int yc = min(y_t + n, pl_h - 1);
                const unsigned ra = (unsigned)(src_y0 + yc) * src_stride32 + (unsigned)(src_x0 + xc) * 3u;
                const unsigned al = src_lo2 + ra;
                sh[n] = (int)(al & 3u);
                const int dmax = (int)((((al + nb - 1u) & ~3u) - (al & ~3u)) >> 2);
                const unsigned vo = (ra - (unsigned)sh[n]) + 4u * (unsigned)min(lane & 15, dmax);
                if ((lane >> 4) == n) voff = vo;
            }
            if (active && lane < 32) glds4_s(a.src_u8, voff, resid_lds);
        }
        f32x4 acc[2][MB];
        {
            // ---- k-loop phase: per (dx, ch) column of taps, halo rows Rr = 0..3: the fragment of row Rr
            // meets the weights of tap row dy = Rr (output row 0) and of dy = Rr - 1 (output row 1, the
            // A fragments kept from the previous step): 24 B reads + 54 A reads per 108 MFMAs -------
            __builtin_amdgcn_s_setprio(2);
            const char* buf = smem + cur * SLOTB;
            const char* bbase = buf + ((2 * rp) * PW + 16 * cc + (lane & 15)) * G::LPIXB + (lane >> 4) * 16;
            auto read_b = [&](int st) __attribute__((always_inline)) -> half8 {
                const int Rr = st & 3, ch = (st >> 2) & 1, dx = st >> 3;
                return *(const half8*)(bbase + (Rr * PW + dx) * G::LPIXB + ch * 64);
            };
            auto read_a = [&](int st, int m) __attribute__((always_inline)) -> half8 {   // weights of tap (dy = Rr, dx), half ch
                const int Rr = st & 3, ch = (st >> 2) & 1, dx = st >> 3;
                return *(const half8*)(wrd + ((((Rr * 3 + dx) * 2 + ch) * MB + m) * 1024));
            };
            constexpr int RQ = PFF + 1;
            constexpr int AD = TAIL4_AD, AQ = AD + 2;     // weight fragments read AD steps ahead; steps st and st - 1 are in use
            half8 bq[RQ];
            half8 aq[AQ][MB];
            __builtin_amdgcn_sched_barrier(0);
#pragma unroll
            for (int f = 0; f < PFF; ++f) bq[f] = read_b(f);
#pragma unroll
            for (int f = 0; f < AD; ++f)
#pragma unroll
                for (int m = 0; m < MB; ++m) aq[f][m] = read_a(f, m);
            __builtin_amdgcn_sched_barrier(0);
            // pinned step by step like the 2x tail's (left alone hipcc waits for every fragment right in front of its MFMAs)
            static_for<NSTEP>([&](auto sc) __attribute__((always_inline)) {
                constexpr int st = decltype(sc)::value, Rr = st & 3;
                if constexpr (st + PFF < NSTEP) bq[(st + PFF) % RQ] = read_b(st + PFF);
                if constexpr (st + AD < NSTEP && ((st + AD) & 3) <= 2) {
#pragma unroll
                    for (int m = 0; m < MB; ++m) aq[(st + AD) % AQ][m] = read_a(st + AD, m);
                }
                const half8 b = bq[st % RQ];
#pragma unroll
                for (int m = 0; m < MB; ++m) {
                    if constexpr (Rr <= 2)    // output row 0, tap (dy = Rr, dx)
                        acc[0][m] = __builtin_amdgcn_mfma_f32_16x16x32_f16(aq[st % AQ][m], b, st == 0 ? zero4 : acc[0][m], 0, 0, 0);
                    if constexpr (Rr >= 1)    // output row 1, tap (dy = Rr - 1, dx): the previous step's weights
                        acc[1][m] = __builtin_amdgcn_mfma_f32_16x16x32_f16(aq[(st + AQ - 1) % AQ][m], b, st == 1 ? zero4 : acc[1][m], 0, 0, 0);
                }
                __builtin_amdgcn_sched_barrier(0);
            });
            __builtin_amdgcn_s_setprio(0);
        }
        // nothing was issued to the memory pipe during the k-loop except the residual dwords at its
        // top: waiting for everything is cheap, and proves tile k+1 (refilled one full phase ago by the
        // other group) for that group's k-loop, which starts now
        tile_barrier<0>();
        // ---- epilogue phase ----------------------------------------------------------------------
        const Sched la = read_sched(min(k + TRUNK_LOOKAHEAD, nsched - 1));   // tile k+3 goes into the slot just consumed
        if (active) {
            const int lane_o = opaque(lane);
            const int g = lane_o >> 4, p = lane_o & 15;          // sub-row of the 4x4 output block, pixel
            const float norm = (float)(1 / 255.0);
            const int core_y0 = ctx.core_y0, core_y1 = min(ctx.core_y1, pl_h);
            const int core_x0 = ctx.core_x0, core_x1 = min(ctx.core_x1, pl_w);
            const bool col_ok = xs + p >= core_x0 && xs + p < core_x1;
            const bool aligned = (((size_t)a.dst_u8 | a.dst_stride) & 3) == 0;
#pragma unroll
            for (int n = 0; n < 2; ++n) {
                const int y = y_t + n;
                // (value + bias + residual) * 255 as floats; v_cvt_pk_u8_f32 rounds half to even, saturates and drops the byte
                // into its place: cv2's convertTo(CV_8U) and the packing in one instruction per byte
                float q[MB][4];
                unsigned rb[MB];
#pragma unroll
                for (int m = 0; m < MB; ++m) rb[m] = *(const uint8_t*)(resid_rd + n * 64 + sh[n] + 3 * p + m);
#pragma unroll
                for (int m = 0; m < MB; ++m) {
                    const float res = (float)rb[m] * norm;
#pragma unroll
                    for (int j = 0; j < 4; ++j) q[m][j] = ((acc[n][m][j] + b4[m][j]) + res) * 255.0f;
                }
                if (y >= core_y0 && y < core_y1 && col_ok) {
                    // output row 4y + g, pixels 4(x) .. 4(x)+3, BGR each: 12 contiguous bytes
                    uint8_t* d = a.dst_u8 + ((size_t)(src_y0 + y) * R + g) * a.dst_stride + (size_t)(src_x0 + xs + p) * R * 3;
                    auto pk4 = [](float b0, float b1, float b2, float b3) __attribute__((always_inline)) {
                        unsigned r = __builtin_amdgcn_cvt_pk_u8_f32(b0, 0, 0u);
                        r = __builtin_amdgcn_cvt_pk_u8_f32(b1, 1, r);
                        r = __builtin_amdgcn_cvt_pk_u8_f32(b2, 2, r);
                        return __builtin_amdgcn_cvt_pk_u8_f32(b3, 3, r);
                    };
                    const unsigned d0 = pk4(q[0][0], q[1][0], q[2][0], q[0][1]);
                    const unsigned d1 = pk4(q[1][1], q[2][1], q[0][2], q[1][2]);
                    const unsigned d2 = pk4(q[2][2], q[0][3], q[1][3], q[2][3]);
                    if (aligned) {
                        ((unsigned*)d)[0] = d0; ((unsigned*)d)[1] = d1; ((unsigned*)d)[2] = d2;
                    } else {
#pragma unroll
                        for (int e = 0; e < 4; ++e) {
                            d[e] = (uint8_t)(d0 >> (8 * e)); d[4 + e] = (uint8_t)(d1 >> (8 * e)); d[8 + e] = (uint8_t)(d2 >> (8 * e));
                        }
                    }
                }
            }
        }
        // refill the slot this group has just consumed with tile k+3, behind the stores in the queue
#pragma unroll
        for (int i = 0; i < CPW; ++i) trunk_issue_piece<NF>(la.base, la.pitch, lds0 + cur * SLOTB, i, wave, dma_pc[i]);
        ctx = load_ctx(min(k + 2, nsched - 1));           // the group's next tile (past the end: an inert entry)
        group_barrier();
        cur = cur + 2 >= TAIL4_SLOTS ? cur + 2 - TAIL4_SLOTS : cur + 2;
    }
    if (grp == 0) group_barrier();
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
}

// ----------------------------------------------------------------------------------------------
// head_kernel<NF, SRC>: from_pixels(PIXEL_BGR) + substract_mean_normalize + Conv_0 (3 -> NF) +
// PReLU_1, fp16 NHWC out.  SRC 0: u8 HWC source; the integer pixel values go through the MFMA
// exactly (0..255 are exact in fp16) and the 1/255 normalisation is applied to the fp32
// accumulator.  SRC 1: f32 planar source (an ncnn::Mat the caller normalised), rounded to fp16.
// K is laid out as [tap][4] (3 channels + 1 zero) -> 36, padded to 3 k-steps of 16.
// ----------------------------------------------------------------------------------------------
#define HEAD_WPE 3            // waves per SIMD the register allocation aims at (= workgroups per CU: one wave per SIMD each)
template <int NF, int SRC>
__global__ __launch_bounds__(256, HEAD_WPE) void head_kernel(HeadArgs a)
{
    constexpr int MF = (NF + 31) / 32;
    __shared__ __attribute__((aligned(16))) char hsm[NPIX * 8 + PARAM_LDS + 4 * StageGeo<NF>::BYTES];
    half4* tile = (half4*)hsm;
    float* bias_lds = (float*)(hsm + NPIX * 8);
    float* slope_lds = bias_lds + 64;

    const int wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
    const int lane = threadIdx.x & 63;
    const int half = lane >> 5;
    const int px = lane & 31;

    // Everything a workgroup needs from memory is requested up front -- weights, parameters, the plane lookup -- and
    // the tile's pixels in ONE batch behind the lookup: two dependent round trips per workgroup instead of four
    // (weights were fetched behind the pixel loop, whose iterations each waited for their own loads).  The kernel
    // writes 128 B per pixel and computes next to nothing: its time is the length of this chain divided by the
    // workgroups a CU holds (profiles/r04_ab_results.txt block 17).
    half8 w[3][MF];
#pragma unroll
    for (int ks = 0; ks < 3; ++ks)
#pragma unroll
        for (int m = 0; m < MF; ++m) w[ks][m] = a.wpk[(ks * MF + m) * 64 + lane];
    float prm_b = 0.f, prm_s = 0.f;
    if (threadIdx.x < MF * 32) { prm_b = a.bias[threadIdx.x]; prm_s = a.slope[threadIdx.x]; }
    const PlaneDesc* gpl = a.planes;
    TileId id;
    id.plane = __builtin_amdgcn_readfirstlane(
        find_plane([gpl](int i) { return gpl[i].tile_begin; }, a.nplanes, (int)blockIdx.x, lane));
    const PlaneDesc& pl = a.planes[id.plane];
    const int pl_h = __builtin_amdgcn_readfirstlane(pl.h), pl_w = __builtin_amdgcn_readfirstlane(pl.w);
    const int src_y0 = __builtin_amdgcn_readfirstlane(pl.src_y0), src_x0 = __builtin_amdgcn_readfirstlane(pl.src_x0);
    {
        const int local = (int)blockIdx.x - __builtin_amdgcn_readfirstlane(pl.tile_begin);
        const int ntx = __builtin_amdgcn_readfirstlane(pl.ntx);
        id.ty = local / ntx;
        id.tx = local - id.ty * ntx;
    }
    // the halo tile's pixels: 340 for 256 threads -- two per thread, both requested before either is used
    constexpr int PPT = (NPIX + 255) / 256;
    float pv[PPT][3];
    bool pin[PPT];
#pragma unroll
    for (int k = 0; k < PPT; ++k) {
        // loads without a condition around them (addresses clamped into the plane, the value dropped afterwards): inside an
        // exec-masked block hipcc waits for each pixel's bytes before it requests the next pixel's
        const int p = min((int)threadIdx.x + 256 * k, NPIX - 1);
        const int r = p / PW, c = p - (p / PW) * PW;
        const int y = id.ty * TH + r - 1, x = id.tx * TW + c - 1;
        pin[k] = y >= 0 && y < pl_h && x >= 0 && x < pl_w;      // zero padding at the PLANE edge
        const int yc = min(max(y, 0), pl_h - 1), xc = min(max(x, 0), pl_w - 1);
        if constexpr (SRC == 0) {
            const uint8_t* sp = a.src_u8 + (size_t)(src_y0 + yc) * a.src_stride + (size_t)(src_x0 + xc) * 3;
            pv[k][0] = (float)sp[0]; pv[k][1] = (float)sp[1]; pv[k][2] = (float)sp[2];
        } else {
            const size_t hw = (size_t)pl_h * pl_w, o = (size_t)yc * pl_w + xc;
            pv[k][0] = a.src_f32[o]; pv[k][1] = a.src_f32[hw + o]; pv[k][2] = a.src_f32[2 * hw + o];
        }
    }
    if (threadIdx.x < 64) {
        bias_lds[threadIdx.x] = prm_b;
        slope_lds[threadIdx.x] = prm_s;
        slope_lds[64 + threadIdx.x] = prm_s <= 1.f ? __builtin_inff() : -__builtin_inff();
    }
#pragma unroll
    for (int k = 0; k < PPT; ++k) {
        const int p = threadIdx.x + 256 * k;
        if (p < NPIX) {
            half4 v = {(_Float16)pv[k][0], (_Float16)pv[k][1], (_Float16)pv[k][2], (_Float16)0.f};
            if (!pin[k]) v = half4{0, 0, 0, 0};
            tile[p] = v;
        }
    }
    __syncthreads();

    f32x16 acc[2][MF];
#pragma unroll
    for (int n = 0; n < 2; ++n) {
#pragma unroll
        for (int m = 0; m < MF; ++m)
#pragma unroll
            for (int r = 0; r < 16; ++r) acc[n][m][r] = 0.f;
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
                acc[n][m] = __builtin_amdgcn_mfma_f32_32x32x16_f16(w[ks][m], b, acc[n][m], 0, 0, 0);
        }
#pragma unroll
        for (int m = 0; m < MF; ++m)
#pragma unroll
            for (int g = 0; g < 4; ++g) {
                const f32x4 b4 = *(const f32x4*)(bias_lds + 32 * m + 8 * g + 4 * half);
#pragma unroll
                for (int j = 0; j < 4; ++j) acc[n][m][4 * g + j] = acc[n][m][4 * g + j] * a.in_scale + b4[j];
            }
    }
    store_trunk_rows<NF, MF>(acc, slope_lds, hsm + NPIX * 8 + PARAM_LDS + wave * StageGeo<NF>::BYTES, a.out_act, pl,
                             id.ty * TH + 2 * wave, id.tx * TW, lane);
}

// ----------------------------------------------------------------------------------------------
// headp_kernel<NF, SRC>: head_kernel's arithmetic (same K layout, same rounding points, same bytes) as a PERSISTENT,
// software-pipelined kernel.  head_kernel writes 128 B per pixel and computes next to nothing, yet ran at half the
// rate a plain store loop reaches (tools/hbm_stream_bench.hip: 273 MB in 39 us): a workgroup lived for its chain of
// dependent loads -- plane lookup, pixels, (weights) -- and a CU holds three of them.  Here a workgroup keeps its
// weights, parameters and the plane table, walks tiles blockIdx.x, + gridDim.x, ... and requests tile t + 1's pixels
// before it computes tile t, so the only thing it waits for is the barrier between "pixels in LDS" and the MFMAs.
// Every store is unconditional (lanes that own nothing write to the sink): a fixed number of memory operations per
// tile, none of them inside a branch.
// ----------------------------------------------------------------------------------------------
constexpr int HEADP_WG_PER_CU = 3;
template <int NF>
constexpr int headp_lds_bytes() { return 2 * NPIX * 8 + PARAM_LDS + PLANE_LDS + 4 * StageGeo<NF>::BYTES; }

template <int NF, int SRC>
__global__ __launch_bounds__(256, HEADP_WG_PER_CU) void headp_kernel(HeadArgs a)
{
    constexpr int MF = (NF + 31) / 32;
    constexpr int SPX = StageGeo<NF>::SPX, PIXB = NF * 2, SPP = NF / 8;
    extern __shared__ __attribute__((aligned(16))) char hsm[];
    half4* const tile0 = (half4*)hsm;
    float* const bias_lds = (float*)(hsm + 2 * NPIX * 8);
    float* const slope_lds = bias_lds + 64;
    PlaneDesc* const planes_lds = (PlaneDesc*)(hsm + 2 * NPIX * 8 + PARAM_LDS);
    int* const tbegin_lds = (int*)(hsm + 2 * NPIX * 8 + PARAM_LDS + MAX_PLANES * 64);
    char* const stage = hsm + 2 * NPIX * 8 + PARAM_LDS + PLANE_LDS + (threadIdx.x >> 6) * StageGeo<NF>::BYTES;

    const int wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
    const int lane = threadIdx.x & 63;
    const int half = lane >> 5;
    const int px = lane & 31;

    half8 w[3][MF];
#pragma unroll
    for (int ks = 0; ks < 3; ++ks)
#pragma unroll
        for (int m = 0; m < MF; ++m) w[ks][m] = a.wpk[(ks * MF + m) * 64 + lane];
    if (threadIdx.x < 64) {
        const float b = threadIdx.x < MF * 32 ? a.bias[threadIdx.x] : 0.f;
        const float sl = threadIdx.x < MF * 32 ? a.slope[threadIdx.x] : 0.f;
        bias_lds[threadIdx.x] = b;
        slope_lds[threadIdx.x] = sl;
        slope_lds[64 + threadIdx.x] = sl <= 1.f ? __builtin_inff() : -__builtin_inff();
    }
    for (int i = threadIdx.x; i < a.nplanes * 16; i += 256) ((int*)planes_lds)[i] = ((const int*)a.planes)[i];
    if ((int)threadIdx.x < a.nplanes) tbegin_lds[threadIdx.x] = a.planes[threadIdx.x].tile_begin;
    __syncthreads();
    PlaneTable pt;
    pt.pl = planes_lds;
    pt.tile_begin = tbegin_lds;
    pt.nplanes = a.nplanes;

    // this thread's (up to) two pixels of a halo tile: position p = row r, column c
    constexpr int PPT = (NPIX + 255) / 256;
    int pr[PPT], pc[PPT];
#pragma unroll
    for (int k = 0; k < PPT; ++k) {
        const int p = min((int)threadIdx.x + 256 * k, NPIX - 1);
        pr[k] = p / PW;
        pc[k] = p - pr[k] * PW;
    }
    struct Tile { int plane, ty, tx, pl_h, pl_w, src_y0, src_x0; };
    auto locate = [&](int t) __attribute__((always_inline)) {
        const TileId id = pt.decode(t, lane);
        Tile r;
        r.plane = id.plane; r.ty = id.ty; r.tx = id.tx;
        const PlaneDesc& pl = planes_lds[id.plane];
        r.pl_h = __builtin_amdgcn_readfirstlane(pl.h); r.pl_w = __builtin_amdgcn_readfirstlane(pl.w);
        r.src_y0 = __builtin_amdgcn_readfirstlane(pl.src_y0); r.src_x0 = __builtin_amdgcn_readfirstlane(pl.src_x0);
        return r;
    };
    float pv[PPT][3];
    bool pin[PPT];
    // requests the tile's pixels: no condition around the loads (addresses clamped into the plane, the value dropped later)
    auto fetch = [&](const Tile& T) __attribute__((always_inline)) {
#pragma unroll
        for (int k = 0; k < PPT; ++k) {
            const int y = T.ty * TH + pr[k] - 1, x = T.tx * TW + pc[k] - 1;
            pin[k] = y >= 0 && y < T.pl_h && x >= 0 && x < T.pl_w;      // zero padding at the PLANE edge
            const int yc = min(max(y, 0), T.pl_h - 1), xc = min(max(x, 0), T.pl_w - 1);
            if constexpr (SRC == 0) {
                const uint8_t* sp = a.src_u8 + (size_t)(T.src_y0 + yc) * a.src_stride + (size_t)(T.src_x0 + xc) * 3;
                pv[k][0] = (float)sp[0]; pv[k][1] = (float)sp[1]; pv[k][2] = (float)sp[2];
            } else {
                const size_t hw = (size_t)T.pl_h * T.pl_w, o = (size_t)yc * T.pl_w + xc;
                pv[k][0] = a.src_f32[o]; pv[k][1] = a.src_f32[hw + o]; pv[k][2] = a.src_f32[2 * hw + o];
            }
        }
    };

    int t = blockIdx.x;
    if (t >= a.ntiles) return;
    Tile cur = locate(t);
    fetch(cur);
    int buf = 0;
    char* const sink = (char*)a.sink + lane * 16;
    for (; t < a.ntiles; t += gridDim.x) {
        half4* const tile = tile0 + buf * NPIX;
#pragma unroll
        for (int k = 0; k < PPT; ++k) {
            const int p = threadIdx.x + 256 * k;
            if (p < NPIX) {
                half4 v = {(_Float16)pv[k][0], (_Float16)pv[k][1], (_Float16)pv[k][2], (_Float16)0.f};
                if (!pin[k]) v = half4{0, 0, 0, 0};
                tile[p] = v;
            }
        }
        __syncthreads();
        // the next tile's pixels travel while this one is computed (past the end: this tile's again, dropped)
        const int tn = t + (int)gridDim.x < a.ntiles ? t + (int)gridDim.x : t;
        const Tile nxt = locate(tn);
        fetch(nxt);

        f32x16 acc[2][MF];
#pragma unroll
        for (int n = 0; n < 2; ++n) {
#pragma unroll
            for (int m = 0; m < MF; ++m)
#pragma unroll
                for (int r = 0; r < 16; ++r) acc[n][m][r] = 0.f;
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
                    acc[n][m] = __builtin_amdgcn_mfma_f32_32x32x16_f16(w[ks][m], b, acc[n][m], 0, 0, 0);
            }
        }
        // bias, PReLU (med3 form, as store_trunk_rows), fp16 -> the wave's staging area.  The lane's parameters are read once
        // per channel group for both rows; scale + bias and the slope product on float pairs (v_pk_fma_f32, v_pk_mul_f32: one
        // rounding each, as the scalar forms)
        const f32x2 sc2 = {a.in_scale, a.in_scale};
#pragma unroll
        for (int m = 0; m < MF; ++m) {
#pragma unroll
            for (int g = 0; g < 4; ++g) {
                if (32 * m + 8 * g >= NF) continue;   // NF is a multiple of 8: groups are all-or-nothing
                const int cb = 32 * m + 8 * g + 4 * half;
                const f32x4 b4 = *(const f32x4*)(bias_lds + cb);
                const f32x4 s4 = *(const f32x4*)(slope_lds + cb);
                const f32x4 i4 = *(const f32x4*)(slope_lds + 64 + cb);
#pragma unroll
                for (int n = 0; n < 2; ++n) {
                    const f32x2 x01 = __builtin_elementwise_fma(f32x2{acc[n][m][4 * g], acc[n][m][4 * g + 1]}, sc2, f32x2{b4[0], b4[1]});
                    const f32x2 x23 = __builtin_elementwise_fma(f32x2{acc[n][m][4 * g + 2], acc[n][m][4 * g + 3]}, sc2, f32x2{b4[2], b4[3]});
                    const f32x2 t01 = x01 * f32x2{s4[0], s4[1]}, t23 = x23 * f32x2{s4[2], s4[3]};
                    const f32x2 v01 = {__builtin_amdgcn_fmed3f(x01[0], t01[0], i4[0]), __builtin_amdgcn_fmed3f(x01[1], t01[1], i4[1])};
                    const f32x2 v23 = {__builtin_amdgcn_fmed3f(x23[0], t23[0], i4[2]), __builtin_amdgcn_fmed3f(x23[1], t23[1], i4[3])};
                    uint2 o;
                    o.x = __builtin_bit_cast(unsigned, __builtin_convertvector(v01, half2v));
                    o.y = __builtin_bit_cast(unsigned, __builtin_convertvector(v23, half2v));
                    *(uint2*)(stage + (n * 32 + px) * SPX + cb * 2) = o;
                }
            }
        }
        __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
        __builtin_amdgcn_wave_barrier();
        __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
        {
            const PlaneDesc& pl = planes_lds[cur.plane];
            const int pitch = __builtin_amdgcn_readfirstlane(pl.pitch);
            const long long act_off = ((long long)__builtin_amdgcn_readfirstlane((int)(pl.act_off >> 32)) << 32) |
                                      (unsigned)__builtin_amdgcn_readfirstlane((int)pl.act_off);
            const int y0 = cur.ty * TH + 2 * wave, x0 = cur.tx * TW;
            const int vx = min(TW, cur.pl_w - x0);          // valid pixels of this tile row (uniform)
            constexpr int NIT = (32 * SPP + 63) / 64;
#pragma unroll
            for (int n = 0; n < 2; ++n) {
                const int y = y0 + n;
                char* const grow = (char*)a.out_act + ((size_t)act_off + (size_t)(y + 1) * pitch + (x0 + 1)) * PIXB;
#pragma unroll
                for (int k = 0; k < NIT; ++k) {
                    const int q = min(k * 64 + lane, 32 * SPP - 1);
                    const int pix = q / SPP, slot = q - pix * SPP;
                    const bool ok = y < cur.pl_h && k * 64 + lane < 32 * SPP && pix < vx;
                    char* const dst = ok ? grow + pix * PIXB + slot * 16 : sink;
                    *(uint4*)dst = *(const uint4*)(stage + (n * 32 + pix) * SPX + slot * 16);
                }
            }
        }
        cur = nxt;
        buf ^= 1;
    }
}

}  // namespace uva
