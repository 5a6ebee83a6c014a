// uva_wino.hip.h -- trunkw_kernel<64>: TWO consecutive 64 -> 64 trunk layers (each + bias + PReLU) per launch,
// like trunk2_kernel, but every 3x3 convolution is evaluated as 1-D WINOGRAD F(2,3) ALONG x: four multiplications
// for two output columns instead of six, i.e. 96 instead of 144 v_mfma_f32_16x16x32_f16 per wave and k-loop.
// (reference graph: models/2x_Compact_Pretrain.param:7-37, models/4x_Compact_Pretrain.param:7-37 -- the sixteen
// Convolution 64->64 3x3 pad 1 + PReLU layers; ncnn itself runs 3x3 stride-1 convolutions through Winograd
// transforms, src/layer/vulkan/convolution_vulkan.cpp.)
//
// Why: trunk2_kernel runs at the package power limit, and the matrix pipes are most of that energy
// (profiles/r04_ab_results.txt block 1: dropping a third of its MFMAs, same fragment reads, takes the launch from
// 268 to 220 us on the same box; doing the input transform redundantly at fragment-read time gives half of that
// back -- a v_pk_add_f16 costs about a tenth of an MFMA).  So the transform is done ONCE per value and the
// transformed values live in LDS:
//
//   output columns come in PAIRS (xa, xa+1); with d0..d3 the input at columns xa-1..xa+2 and g0,g1,g2 a filter row:
//     V0 = d0 - d2   V1 = d1 + d2   V2 = d2 - d1   V3 = d1 - d3                (fp16, one rounding each)
//     U0 = g0   U1 = (g0+g1+g2)/2   U2 = (g0-g1+g2)/2   U3 = g2               (fp16, host: pack_trunk64_wino)
//     Mj = sum over input channels and filter rows of Uj * Vj                   (MFMA, fp32 accumulate)
//     out(xa) = M0 + M1 + M2,   out(xa+1) = M1 - M2 - M3                        (fp32, then bias, PReLU, fp16)
//   The CPU checker restates exactly this, rounding points included (conv2d_wino_f23, mode UVO_WINOGRAD_F23; test side).
//
// Organisation (one persistent 8-wave workgroup per CU, walking DOWN a 30-column strip of a plane in 4-row steps):
//   group A (waves 0-3) = layer i, group B (waves 4-7) = layer i+1, ping-pong as in trunk2_kernel: A's k-loop runs
//   beside B's epilogue (HBM stores) in phase X, B's k-loop beside A's epilogue in phase Y.  A wave owns 16 OUTPUT
//   CHANNELS (96 weight registers: 4 transformed taps x 3 rows x 64 input channels) and all 4 rows x 16 pairs of a
//   step: 16 accumulators [row][j], 48 fragment reads and 96 MFMAs per k-loop, every fragment used by up to three
//   filter rows.  Both k-loops read TRANSFORMED rows:
//     A-ring (6 rows):  filled in phase Y from the 4 new RAW input rows of the next step (LDS-DMA by A's waves into a
//                       2-slot staging area one and a half periods ahead; B's wave w transforms row w once its k-loop
//                       is done -- A's epilogue is the longer half of that phase); the two rows a step shares with
//                       the previous one stay where they are;
//     B-ring (10 rows): A's epilogue turns its result pairs into V0..V3 with the neighbouring lane's pair
//                       (v_mov_dpp row_shl:1) and writes them here; B is 1.5 steps behind.
//   Pair p of a step's block: A computes intermediate columns x0-1+2p, x0+2p (16 pairs = 32 columns from the 34
//   input columns x0-2 .. x0+31), B output columns x0+2p, x0+2p+1 (15 pairs from A's 32 columns).
//   A segment (a run of steps down one strip) starts WITHOUT the two shared rows: the first block's first two
//   intermediate rows are garbage and its consumer step is idle -- k steps yield 4(k-1) output rows (host:
//   build_trunkw_schedule; the CPU test walks the lists and checks that every pixel is produced exactly once from
//   valid rows).
//
// LDS (162 304 of 163 840 bytes):
//   transformed rows: [j 0..3][channel half][pair][4 x 16 B], unit o of pair p in slot o ^ ((p >> 1) & 3):
//     fragment reads (lane (o, p)) and the 16-byte writes (8 consecutive pairs) are bank-conflict free;
//   raw rows: 34 pixel records of 128 B, column cc = 2h + par in record h + 17 par, octet in slot octet ^ (h & 7):
//     the transform's reads of columns 2p + k are conflict free (tools/wino_swizzle_check.py).
#pragma once
#include <type_traits>

#include "uva_devutil.hip.h"
#include "uva_wino.h"

namespace uva {

constexpr int TW_AROWB = 4 * 2 * 16 * 64;         // 8192: one transformed row of the A-ring (16 pairs)
constexpr int TW_BROWB = 4 * 2 * 15 * 64;         // 7680: one transformed row of the B-ring (15 pairs)
constexpr int TW_AROWS = 6, TW_BROWS = 10;
constexpr int TW_RAWROWB = 34 * 128;              // 4352
constexpr int TW_RAWSLOTB = 4 * TW_RAWROWB;       // 17 408 = 17 LDS-DMA pieces
constexpr int TW_RAW_PIECES = TW_RAWSLOTB / 1024;
constexpr int TW_ARING = 0;
constexpr int TW_BRING = TW_ARING + TW_AROWS * TW_AROWB;
constexpr int TW_RAW = TW_BRING + TW_BROWS * TW_BROWB;
constexpr int TW_PRM = TW_RAW + 2 * TW_RAWSLOTB;
constexpr int TW_FLAGS_OFF = TW_PRM + 2 * PARAM_LDS + 64;       // (TW_FLAGS) two step counters: [0] producers' k-loops done, [1] consumers'
constexpr int TW_LDS_BYTES = TW_FLAGS_OFF + 16;                 // + a spare unit per lane group (pair 15's writes) + the counters
static_assert(TW_RAW_PIECES == 17 && TW_RAWSLOTB % 1024 == 0, "raw slot = whole LDS-DMA pieces");
static_assert(TW_LDS_BYTES <= 160 * 1024, "trunkw kernel LDS budget");
static_assert(TW_RAW % 128 == 0 && TW_RAWROWB % 128 == 0, "raw pixel records are 128-byte aligned (the octet XOR flips address bits 4..6)");

// Step g of a workgroup (host: build_trunkw_schedule).  Trunk2Step's 32 bytes, other meaning:
//   a: x = byte offset (low 32) of input pixel (yA + 1, x0 - 2) -- the first of the step's four NEW input rows --
//      from the activation buffer's base (guard included), y = offset bits 32..39 | row mask << 8 (bit n:
//      intermediate row yA + n is inside the plane) | c_lo << 12 | c_hi << 18 (intermediate columns [c_lo, c_hi) of
//      the 32 are inside the plane) | active << 24, z = row pitch in bytes
//      active << 24 | (the workgroup's first step only) all six input rows are to be fetched << 25, z = row pitch in bytes
//   b: x = byte offset (low 32) of output pixel (yA - 1, x0), y = offset bits 32..39 | v1 << 8 | valid columns << 11 |
//      v0 << 17 (rows yA-1 + [v0, v1) are stored) | active << 24, z = row pitch in bytes
//   FOLDED steps (a.y bit 27, b.y bit 25; host: build_trunkw_schedule): the last, narrow (<= TW_FOLD_MAXW = 12 columns) strips of TWO planes of one
//   size in one walk -- pairs 0..7 belong to the first plane, pairs 8..15 to the second, whose pixels lie a constant further on:
//   .w = that constant - 2048 (raw column 16 + c of the step is the second plane's column c: 16 pixels back, one plane on), the
//   first plane's index in the top byte of .z (the pitch is its low 24 bits in every entry).  Rings,
//   transforms and k-loops are pair-wise and unchanged; the DMA source, the producers' column masks (pair & 7) and the consumers'
//   store addresses are what a folded step does differently.  Pair 7 of either half mixes the two planes (its raw columns 16, 17
//   are the other plane's) and is never used: a folded strip has at most 12 columns = 6 consumer pairs, which need producer pairs 0..6.

// LDS-DMA piece i of wave `wave`: piece c = 4i + wave covers units [64c, 64c + 64) of a raw slot; unit q is row
// q / 272, record (q % 272) / 8, slot q % 8 -> (row << 13) | byte offset of that octet inside the source row
__device__ __forceinline__ unsigned tw_piece_const(int i, int wave, int lane)
{
    const int q = (4 * i + wave) * 64 + lane;
    const int r = q / 272, u = q - r * 272;
    const int rec = u >> 3, sl = u & 7;
    const int par = rec >= 17 ? 1 : 0, h = rec - 17 * par;
    return (unsigned)((r << 13) | ((2 * h + par) * 128 + ((sl ^ (h & 7)) << 4)));
}

__device__ __forceinline__ unsigned dpp_row_shl1(unsigned v)
{
    return (unsigned)__builtin_amdgcn_mov_dpp((int)v, 0x101, 0xf, 0xf, true);    // lane p <- lane p + 1 of its row of 16 (lane 15: 0)
}
__device__ __forceinline__ unsigned pk_add_f16(unsigned a, unsigned b)
{
    return __builtin_bit_cast(unsigned, (half2v)(__builtin_bit_cast(half2v, a) + __builtin_bit_cast(half2v, b)));
}

// wave priorities inside the k-loops / everywhere else (A/B builds override them)
#define TW_PRIO_K 2
#define TW_PRIO_E 0
// per group (A = waves 0-3, the producers; B = waves 4-7, the consumers, whose phases are the longer ones): k-loop / elsewhere.
// The guide's "static priority for the younger half" is TW_PRIO_KB = TW_PRIO_EB = 1 with group A at 0 throughout.
#define TW_PRIO_KA TW_PRIO_K
#define TW_PRIO_EA TW_PRIO_E
#define TW_PRIO_KB TW_PRIO_K
#define TW_PRIO_EB TW_PRIO_E

#ifdef UVA_INSTRUMENT
#define TW_STAMP(k) do { if (stamp) a.dbg[16 * it + 8 * grp + (k)] = __builtin_amdgcn_s_memtime(); } while (0)
#else
#define TW_STAMP(k) do { } while (0)
#endif

template <int NF, int ACT>
__global__ __launch_bounds__(512, 2) UVA_NO_PK_F32 void trunkw_kernel(TrunkwArgs a)
{
    static_assert(NF == 64, "written for 64 features");
    constexpr int PIXB = 128;
    // Schedule constants.  Each is the survivor of a measured A/B (profiles/r04_ab_results.txt, r05_ab_results.txt); the
    // variants that lost -- raw rows transformed by the producers (TW_XA), epilogue rows inside the k-loop (TW_INROWS), LDS step
    // counters instead of barriers (TW_FLAGS), the producers' LDS-DMA issued late (TW_DMA_LATE) or by both groups (TW_DMA_B < 5),
    // deferred producer rows (TW_DEFER_A), both groups pre-reading (TW_PRE_BAR = 3), 128-bit stores (TW_W128) and the two
    // wrong-result ceiling experiments (TW_EXP_2D, TW_ABL_*) -- left this file in round 6: tools/experiments/tw_variants_r05.patch.
    constexpr int TW_PFF = 12;        // fragments the producers' k-loop reads ahead of its MFMAs (an LDS read takes ~280 cycles to come
                                      // back while four waves stream fragments, an MFMA 16; 6 until round 5: +0.3-0.6 %, 254 VGPRs)
    constexpr int TW_PFF_B = 6;       // ... the consumers': the in-stream raw-row transform has the registers (7: no faster; 8 spills; 4: -0.5 %).
                                      // The consumers read these first fragments (window row 0: written a period or more ago) in FRONT of the
                                      // barrier that opens their phase and pass it without waiting for them; the producers read behind theirs.
    // The consumers transform the next step's raw rows INSIDE their k-loop: four ds_read_b128 at fragment TW_RAW_F0, the four V rows of a
    // channel half TW_RAW_GAP fragments later (an LDS round trip), one every TW_RAW_STRIDE fragments, then the other half.  Their serial
    // chain epilogue -> raw rows -> k-loop was what a period was made of: +3.6-3.9 % on three boxes (0 / 14 / 3), +0.6-0.8 % more with
    // the k-loop's own first fragments going first (4 / 12 / 3).
    constexpr int TW_RAW_F0 = 4, TW_RAW_GAP = 12, TW_RAW_STRIDE = 3;
    // Of a wave's five LDS-DMA pieces of step it + 2's raw rows the CONSUMERS issue all, at the top of their phase X (their epilogue was
    // the short side of that phase: ~600 ticks of slack), and wait for what they issued an iteration ago in front of barrier 1: rows
    // requested 1.35 periods ahead arrive in time, rows requested one period ahead do not (the DMA's latency under this load is ~3 us).
    constexpr int TW_DMA_PIECES = 5;
    constexpr int PFF_A = TW_PFF;                 // fragments read ahead of their MFMAs: an LDS read takes ~280 cycles to come back
                                                  // while four waves stream fragments, and an MFMA 16 (profiles/r04_ab_results.txt)

    extern __shared__ __attribute__((aligned(16))) char smem[];
    const unsigned lds0 = lds_offset(smem);
    float* const prm_all = (float*)(smem + TW_PRM);       // per layer: bias[64], slope[64], med3 selector[64]

    const int wave8 = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
    const int grp = wave8 >> 2;   // 0: producer (layer i), 1: consumer (layer i+1)
    const int wave = wave8 & 3;   // = the 16-channel block this wave computes
    const int lane = threadIdx.x & 63;

#ifdef UVA_INSTRUMENT
    const unsigned long long t_entry = __builtin_amdgcn_s_memtime();
#endif
    // The prologue is a chain of memory round trips (7 us of a 245 us launch by in-kernel stamps).  Shortening it -- step count and
    // first entries requested together through asm, the first rows' DMA issued before anything waits for the parameters -- changed
    // the launch time by nothing (profiles/r04_ab_results.txt block 22: a waiting workgroup costs no energy, and energy is what a
    // launch at the package limit is made of); the reordering that needed no asm stays.
    const Trunk2Step* const steps = a.steps + (size_t)blockIdx.x * (a.max_steps + TW_PAD_STEPS);
    auto load_a = [&](int g) __attribute__((always_inline)) { return scalar_load16(&steps[g].a); };
    auto load_b = [&](int g) __attribute__((always_inline)) { return scalar_load16(&steps[g].b); };
    const uint4 e_first = load_a(0), e_second = load_a(1);        // (every workgroup's list has TW_PAD_STEPS entries at least)
    const int nsteps = __builtin_amdgcn_readfirstlane(a.nsteps[blockIdx.x]);
    if (nsteps <= 0) return;
#ifdef UVA_INSTRUMENT
    const bool stamp = a.dbg != nullptr && blockIdx.x == 0 && (threadIdx.x & 255) == 0;
    if (stamp && grp == 0) a.dbg[16 * (nsteps + 2)] = t_entry;
#endif

    float prm_b = 0.f, prm_s = 0.f;
    if (wave == 0) {
        prm_b = (grp ? a.bias[1] : a.bias[0])[lane];
        prm_s = (grp ? a.slope[1] : a.slope[0])[lane];
    }
    // this wave's 16 output channels of its layer's transformed weights, resident for the whole kernel
    half8 w[24];                                  // [(j * 3 + dy) * 2 + ch]
    {
        const half8* wp = (const half8*)(grp ? a.wpk[1] : a.wpk[0]);
#pragma unroll
        for (int i = 0; i < 24; ++i) w[i] = wp[(i * 4 + wave) * 64 + lane];
    }
    // LDS-DMA source position of this lane in piece i (wave 0: five pieces, the others four), two per register
    unsigned dma_pc2[3];
#pragma unroll
    for (int i = 0; i < 3; ++i)
        dma_pc2[i] = tw_piece_const(2 * i, wave, lane) | (2 * i + 1 < 5 ? tw_piece_const(2 * i + 1, wave, lane) << 16 : 0u);
    auto issue_rows = [&](const uint4 e, int slot, auto i0c, auto i1c) __attribute__((always_inline)) {
        constexpr int I0 = decltype(i0c)::value, I1 = decltype(i1c)::value;      // pieces [I0, I1) of this wave's five
        const unsigned ey = __builtin_amdgcn_readfirstlane(e.y);
        const unsigned lo = __builtin_amdgcn_readfirstlane(e.x), hi = ey & 0xffu;
        const char* base = a.in_act + (((unsigned long long)hi << 32) | lo);
        const int pitch = __builtin_amdgcn_readfirstlane(e.z) & 0xffffff;
        // a folded step's raw columns 16.. come from the second plane (uniform branch: one strip in thirty-three)
        const unsigned fadd = ((ey >> 27) & 1u) ? (unsigned)__builtin_amdgcn_readfirstlane(e.w) : 0u;
#pragma unroll
        for (int i = I0; i < I1; ++i) {
            if (4 * i + wave >= TW_RAW_PIECES) continue;
            const unsigned pc = (dma_pc2[i >> 1] >> (16 * (i & 1))) & 0xffffu;
            // (a 24-bit multiply-add: hipcc's v_mad_u64_u32 for the plain expression takes an UNDEFINED register as the high half of
            // its addend, which in the prologue was one a parameter load was still writing -- a wait in front of the first DMA)
            unsigned off = __umul24(pc >> 13, (unsigned)pitch) + (pc & 0x1fffu);
            if (fadd) off += (pc & 0x1800u) ? fadd : 0u;
            glds16_s(base, off, lds0 + TW_RAW + slot * TW_RAWSLOTB + (4 * i + wave) * 1024);
        }
    };

    // lane part of a transformed-row address, fragment / transform view (pair pq = lane & 15, K octet oq = lane >> 4): pair
    // record (64 B) + swizzled unit.  Recomputed from an opaque copy of the lane id where it is used: everything derived
    // from the lane id is loop invariant, and hipcc would keep it all in registers across the k-loops (or spill it)
    auto vlane_of = [](int l) __attribute__((always_inline)) { const int p_ = l & 15, o_ = l >> 4; return (unsigned)(p_ * 64 + ((o_ ^ ((p_ >> 1) & 3)) << 4)); };

    // Raw rows -> transformed rows of the A-ring: wave w transforms the step's new row w.  Lane (oq, pq) handles octets
    // oq and 4 + oq of pair pq: d0..d3 = columns 2pq .. 2pq+3 (records pq / pq+1, parity planes 17 records apart).
    // per-lane constants of the whole kernel (a dozen registers that stay put; recomputing them per step costs more
    // instructions than the kernel can afford: every instruction outside the MFMA stream counts)
    const unsigned vlane_c = vlane_of(lane);
    const unsigned t_lo = (unsigned)((lane & 15) * PIXB + (((lane >> 4) ^ (lane & 7)) << 4));
    const unsigned t_hi = (unsigned)(((lane & 15) + 1) * PIXB + (((lane >> 4) ^ (((lane & 15) + 1) & 7)) << 4));
    auto transform_row = [&](const char* rrow, int pos) __attribute__((always_inline)) {
        const unsigned a_lo = t_lo, a_hi = t_hi;
        char* const vrow = smem + TW_ARING + pos * TW_AROWB + vlane_c;
        half8 d[2][4];
#pragma unroll
        for (int t = 0; t < 2; ++t) {              // all eight reads first: one LDS round trip, not two
            d[t][0] = *(const half8*)(rrow + (a_lo ^ (t ? 64u : 0u)));
            d[t][1] = *(const half8*)(rrow + (a_lo ^ (t ? 64u : 0u)) + 17 * PIXB);
            d[t][2] = *(const half8*)(rrow + (a_hi ^ (t ? 64u : 0u)));
            d[t][3] = *(const half8*)(rrow + (a_hi ^ (t ? 64u : 0u)) + 17 * PIXB);
        }
#pragma unroll
        for (int t = 0; t < 2; ++t) {
            *(half8*)(vrow + 0 * 2048 + t * 1024) = pk_sub(d[t][0], d[t][2]);
            *(half8*)(vrow + 1 * 2048 + t * 1024) = d[t][1] + d[t][2];
            *(half8*)(vrow + 2 * 2048 + t * 1024) = pk_sub(d[t][2], d[t][1]);
            *(half8*)(vrow + 3 * 2048 + t * 1024) = pk_sub(d[t][1], d[t][3]);
        }
    };

    // half a row (octet half t): the unit of work when both groups share a step's four rows (TW_XA = 2)
    auto transform_half = [&](const char* rrow, int pos, int t) __attribute__((always_inline)) {
        const unsigned a_lo = t_lo ^ (t ? 64u : 0u), a_hi = t_hi ^ (t ? 64u : 0u);
        char* const vrow = smem + TW_ARING + pos * TW_AROWB + vlane_c + t * 1024;
        const half8 d0 = *(const half8*)(rrow + a_lo), d1 = *(const half8*)(rrow + a_lo + 17 * PIXB);
        const half8 d2 = *(const half8*)(rrow + a_hi), d3 = *(const half8*)(rrow + a_hi + 17 * PIXB);
        *(half8*)(vrow + 0 * 2048) = pk_sub(d0, d2);
        *(half8*)(vrow + 1 * 2048) = d1 + d2;
        *(half8*)(vrow + 2 * 2048) = pk_sub(d2, d1);
        *(half8*)(vrow + 3 * 2048) = pk_sub(d1, d3);
    };

    auto transform_rows = [&](int slot, int pos0) __attribute__((always_inline)) {
        int pos = pos0 + wave;
        pos = pos >= TW_AROWS ? pos - TW_AROWS : pos;
        transform_row(smem + TW_RAW + slot * TW_RAWSLOTB + wave * TW_RAWROWB, pos);
    };

    // ---- prologue: weights, parameters; A: raw rows of steps 0 and 1, rows of step 0 transformed -----------------
    // The workgroup's first step gets all six input rows (flag in its entry): the two it would share with a step above are
    // fetched by waves 0 and 1 into the (still unused) B-ring and transformed into A-ring rows 0 and 1.
    const bool six = (__builtin_amdgcn_readfirstlane(e_first.y) >> 25) & 1u;
    if (grp == 0) {
        issue_rows(e_first, 0, std::integral_constant<int, 0>{}, std::integral_constant<int, 5>{});
        issue_rows(e_second, 1, std::integral_constant<int, 0>{}, std::integral_constant<int, 5>{});
        if (six && wave < 2) {
            const unsigned lo = __builtin_amdgcn_readfirstlane(e_first.x), hi = __builtin_amdgcn_readfirstlane(e_first.y) & 0xffu;
            const int pitch = __builtin_amdgcn_readfirstlane(e_first.z) & 0xffffff;
            const char* base = a.in_act + (((unsigned long long)hi << 32) | lo) - (size_t)(2 - wave) * pitch;      // rows yA - 1, yA
#pragma unroll
            for (int i = 0; i < 5; ++i) {
                int q = i * 64 + lane;
                q = q < 272 ? q : 271;
                const int rec = q >> 3, sl = q & 7, par = rec >= 17 ? 1 : 0, hh = rec - 17 * par;
                glds16_s(base, (unsigned)((2 * hh + par) * 128 + ((sl ^ (hh & 7)) << 4)), lds0 + TW_BRING + wave * 5120 + i * 1024);
            }
        }
#pragma unroll
        for (int i = 0; i < 24; ++i) asm volatile("" : "+v"(w[i]));
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    }
    if (wave == 0) {                   // (behind the DMA issue: the wait for the parameters runs beside the rows' flight)
        float* prm = prm_all + grp * (PARAM_LDS / 4);
        prm[lane] = prm_b;
        prm[64 + lane] = prm_s;
        prm[128 + lane] = prm_s <= 1.f ? __builtin_inff() : -__builtin_inff();
    }
    group_barrier();                   // every A wave's pieces have landed (each waited for its own)
    if (grp == 1) {
        transform_rows(0, 2);          // (the consumer group fills the A-ring: see its loop)
        if (six && wave < 2) transform_row(smem + TW_BRING + wave * 5120, wave);
    }
    group_barrier();

    const float* const bias_lds = prm_all + grp * (PARAM_LDS / 4);
    const float* const prm_lds = bias_lds + 64;
    const f32x4 zero4 = {0.f, 0.f, 0.f, 0.f};
    // this lane's four output channels' bias: the start value of the M1 accumulators (resident, like the weights)
    const f32x4 bias4 = *(const f32x4*)(bias_lds + 16 * wave + 4 * (lane >> 4));
    // ... and their PReLU slopes and med3 selectors (an LDS read in front of every epilogue is a ~280-cycle stall)
    const f32x4 s4 = *(const f32x4*)(prm_lds + 16 * wave + 4 * (lane >> 4));
    const f32x4 i4 = *(const f32x4*)(prm_lds + 64 + 16 * wave + 4 * (lane >> 4));

    // One k-loop: 48 fragments f = (R * 2 + ch) * 4 + j of 6 transformed rows; fragment (R, j, ch) feeds output rows
    // n = R - dy (dy = 0..2) of accumulator [n][j].  Output row n is complete behind the fragments of window row n + 2; its
    // epilogue is cut into EIGHT SLICES that follow the eight fragments of window row n + 3, in the wave's own instruction
    // stream: slice(n, k), n = 0..2.  There an epilogue instruction costs its wave a few issue cycles; left to the other
    // group's k-loop phase it would get through at a rate of ONE per MFMA of the wave it shares the SIMD with, and the
    // rest would run after that k-loop with nothing beside it (profiles/r04_ab_results.txt).  Row 3 completes with the
    // last MFMA: its slices are the caller's, in the next phase.  The order is pinned fragment by fragment
    // (sched_barrier): the read PFF fragments ahead, this fragment's MFMAs, one slice.
    f32x4 acc[4][4];
    // the first PFF fragments of a k-loop (all of window row 0) into `pre`: TW_PRE_BAR reads them in front of the barrier
    [[maybe_unused]] auto prefetch = [&](auto ring_tag, int base_pos, auto& pre) __attribute__((always_inline)) {
        constexpr bool BR = decltype(ring_tag)::value;
        constexpr int ROWB = BR ? TW_BROWB : TW_AROWB;
        constexpr int PFF = BR ? TW_PFF_B : PFF_A;
        static_assert(PFF <= 8, "the fragments read ahead of the barrier are window row 0's");
        const unsigned r0 = vlane_c + (unsigned)((BR ? TW_BRING : TW_ARING) + base_pos * ROWB);
#pragma unroll
        for (int f = 0; f < PFF; ++f) pre[f] = *(const half8*)(smem + r0 + (f & 3) * (ROWB / 4) + ((f >> 2) & 1) * (ROWB / 8));
    };
    auto kloop = [&](auto ring_tag, int base_pos, auto&& slice, auto&& hook, [[maybe_unused]] const auto& pre) __attribute__((always_inline)) {
        constexpr bool BR = decltype(ring_tag)::value;
        constexpr int INR = 0;             // epilogue rows whose slices ride inside the k-loop (the variant that lost: 0)
        constexpr int ROWB = BR ? TW_BROWB : TW_AROWB, NROWS = BR ? TW_BROWS : TW_AROWS;
        constexpr int JS = ROWB / 4, CS = ROWB / 8;
        constexpr int PFF = BR ? TW_PFF_B : PFF_A;
        constexpr int NFRAG = 48, RQ = PFF + 1;
        unsigned radr[6];
        const unsigned vlane = vlane_c;
#pragma unroll
        for (int R = 0; R < 6; ++R) {
            int pos = base_pos + R;
            pos = pos >= NROWS ? pos - NROWS : pos;
            radr[R] = vlane + (unsigned)((BR ? TW_BRING : TW_ARING) + pos * ROWB);
        }
        auto read_f = [&](int f) __attribute__((always_inline)) -> half8 {
            const int R = f >> 3, ch = (f >> 2) & 1, j = f & 3;
            return *(const half8*)(smem + radr[R] + j * JS + ch * CS);
        };
        half8 bq[RQ];
        __builtin_amdgcn_sched_barrier(0);
#pragma unroll
        for (int f = 0; f < PFF; ++f) {
            if constexpr (BR) bq[f] = pre[f];       // (the consumers: read in front of the barrier)
            else bq[f] = read_f(f);
        }
        __builtin_amdgcn_sched_barrier(0);
        static_for<NFRAG>([&](auto fc) __attribute__((always_inline)) {
            constexpr int f = decltype(fc)::value;
            if constexpr (f + PFF < NFRAG) bq[(f + PFF) % RQ] = read_f(f + PFF);
            constexpr int R = f >> 3, ch = (f >> 2) & 1, j = f & 3;
            const half8 b = bq[f % RQ];
            static_for<4>([&](auto nc) __attribute__((always_inline)) {
                constexpr int n = decltype(nc)::value, dy = R - n;
                if constexpr (dy >= 0 && dy <= 2) {
                    constexpr bool first = dy == 0 && ch == 0;
                    // M1 enters both results with a plus sign: its accumulator starts at the bias
                    acc[n][j] = __builtin_amdgcn_mfma_f32_16x16x32_f16(w[(j * 3 + dy) * 2 + ch], b, first ? (j == 1 ? bias4 : zero4) : acc[n][j], 0, 0, 0);
                }
            });
            if constexpr (R >= 3 && R - 3 < INR) slice(std::integral_constant<int, R - 3>{}, std::integral_constant<int, (f & 7)>{});
            hook(fc);
            __builtin_amdgcn_sched_barrier(0);
        });
    };

    // The slices 0..3 of a row's epilogue, the same for both groups: output transform and PReLU of one block row (the bias
    // is already in M1) -> the pair's two pixels as packed fp16, 4 channels each.  On float pairs: v_pk_add_f32 / v_pk_mul_f32;
    // differences as fma(x, -1, y): one rounding like y - x, and it stays a packed instruction (y - x on float pairs is
    // scalarised into two v_sub_f32, and so is an fma with a visible -1: neg1 is opaque).
    struct RowSt {
        f32x2 u, v;
        unsigned x0, x1, y0, y1;       // x: column 2p (channels 4cg.., 4cg+2..), y: column 2p + 1
        unsigned t[4], q[8];
    };
    f32x2 neg1 = {-1.f, -1.f};
    asm volatile("" : "+v"(neg1));
    auto fin_sum = [&](RowSt& st, int n, int hh) __attribute__((always_inline)) {
        const f32x2 m0 = {acc[n][0][2 * hh], acc[n][0][2 * hh + 1]}, m1 = {acc[n][1][2 * hh], acc[n][1][2 * hh + 1]};
        const f32x2 m2 = {acc[n][2][2 * hh], acc[n][2][2 * hh + 1]}, m3 = {acc[n][3][2 * hh], acc[n][3][2 * hh + 1]};
        st.u = (m0 + m1) + m2;
        st.v = __builtin_elementwise_fma(m3, neg1, __builtin_elementwise_fma(m2, neg1, m1));
    };
    // TW_ACT_F16*: the slopes of the lane's four channels as packed halves
    const half2v s16v[2] = {half2v{(_Float16)s4[0], (_Float16)s4[1]}, half2v{(_Float16)s4[2], (_Float16)s4[3]}};
    auto fin_act = [&](RowSt& st, int hh) __attribute__((always_inline)) {
        if constexpr (ACT != TW_ACT_F32) {
            // (builtins, not inline asm: hipcc has to see these VALU writes to keep the wait states in front of the DPP moves)
            const half2v hu = __builtin_convertvector(st.u, half2v), hv = __builtin_convertvector(st.v, half2v);
            const half2v mu = hu * s16v[hh], mv = hv * s16v[hh];
            const unsigned xu = __builtin_bit_cast(unsigned, __builtin_elementwise_max(hu, mu));
            const unsigned xv = __builtin_bit_cast(unsigned, __builtin_elementwise_max(hv, mv));
            if (hh == 0) { st.x0 = xu; st.y0 = xv; } else { st.x1 = xu; st.y1 = xv; }
            return;
        }
        const f32x2 sl = {s4[2 * hh], s4[2 * hh + 1]};
        const f32x2 us = st.u * sl, vs = st.v * sl;
        const f32x2 pu = {__builtin_amdgcn_fmed3f(st.u[0], us[0], i4[2 * hh]), __builtin_amdgcn_fmed3f(st.u[1], us[1], i4[2 * hh + 1])};
        const f32x2 pv = {__builtin_amdgcn_fmed3f(st.v[0], vs[0], i4[2 * hh]), __builtin_amdgcn_fmed3f(st.v[1], vs[1], i4[2 * hh + 1])};
        const unsigned xu = __builtin_bit_cast(unsigned, __builtin_convertvector(pu, half2v));
        const unsigned xv = __builtin_bit_cast(unsigned, __builtin_convertvector(pv, half2v));
        if (hh == 0) { st.x0 = xu; st.y0 = xv; } else { st.x1 = xu; st.y1 = xv; }
    };

    // The two groups run DISJOINT loops (same number of barriers per iteration): with both roles in one loop body hipcc
    // allocates the two k-loops' accumulators and fragment queues apart and spills the weights.
    const int niter = nsteps + 2;
    // an s_waitcnt hipcc's own bookkeeping can see: without it every k-loop opens with a vmcnt wait for "the weights",
    // i.e. for the LDS-DMA pieces issued in front of it
    __builtin_amdgcn_s_waitcnt(0x0f70);            // vmcnt(0)
    RowSt st2[2];
    if (grp == 0) {
        // ---- group A: k-loop(it) with the epilogue of its rows 0..2 in phase X, row 3 in phase Y ----------------------
        // entries fetched one iteration ahead through the scalar cache: e_own = masks of step it, e_dma = rows of step it + 2 (where the
        // producers would issue pieces of them: they issue none)
        uint4 e_own = load_a(0);
        int a6 = 0;                    // (4 * it) mod 6: A-ring position of the step's first input row
        int b10 = 0;                   // (4 * it) mod 10: B-ring position of the block written in iteration it
        // lane (p, cg) writes units of channel octet 2 wave + (cg >> 1): even cg V0 and V1, odd cg V2 and V3; the lanes of
        // pair 15 (the B-ring holds 15 pairs per plane) write to a spare unit behind the parameters instead: no exec mask
        const int cg = lane >> 4, p = lane & 15;
        const int oo = 2 * (wave & 1) + (cg >> 1);
        const unsigned wlane = p < 15 ? (unsigned)(TW_BRING + (2 * (cg & 1)) * (TW_BROWB / 4) + (wave >> 1) * (TW_BROWB / 8) + p * 64 +
                                                    ((oo ^ ((p >> 1) & 3)) << 4))
                                      : (unsigned)(TW_FLAGS_OFF - 64 + 16 * cg);
        const unsigned w64 = (unsigned)(TW_BRING + (wave >> 1) * (TW_BROWB / 8) + p * 64 + ((oo ^ ((p >> 1) & 3)) << 4) + 8 * (cg & 1));
        const unsigned wrow = p < 15 ? (unsigned)TW_BROWB : 0u;      // (pair 15: every row and both units to the same spare place)
        const unsigned wj = p < 15 ? (unsigned)(TW_BROWB / 4) : 0u;
        half8 pre[PFF_A];
        for (int it = 0; it < niter; ++it) {
            TW_STAMP(0);
            // what the epilogue of a step needs: the masks of a step at its plane's edge, and where its block lies in the B-ring
            // (set for the step whose rows are being finished)
            const unsigned ey = __builtin_amdgcn_readfirstlane(e_own.y);
            int rmask = 15, b10w = b10;
            bool in0 = true, in1 = true, edge = false;
            auto set_step = [&](const unsigned eyv, const int b10v) __attribute__((always_inline)) {
                const int c_lo = (eyv >> 12) & 63, c_hi = (eyv >> 18) & 63;
                rmask = (eyv >> 8) & 15;
                const int pm = ((eyv >> 27) & 1u) ? (p & 7) : p;      // a folded step: each half's pairs count from its own plane's column
                in0 = 2 * pm >= c_lo && 2 * pm < c_hi;
                in1 = 2 * pm + 1 >= c_lo && 2 * pm + 1 < c_hi;
                edge = rmask != 15 || c_lo != 0 || c_hi != 32;
                b10w = b10v;
            };
            auto slice = [&](RowSt& st, auto edge_tag, auto nc, auto kc) __attribute__((always_inline)) {
                constexpr int n = decltype(nc)::value, k = decltype(kc)::value;
                if constexpr (k == 0) fin_sum(st, n, 0);
                else if constexpr (k == 1) fin_act(st, 0);
                else if constexpr (k == 2) fin_sum(st, n, 1);
                else if constexpr (k == 3) fin_act(st, 1);
                else if constexpr (k == 4) {
                    if constexpr (decltype(edge_tag)::value) {            // layer i+1's zero padding
                        const bool rin = (rmask >> n) & 1;
                        const unsigned k0 = (rin && in0) ? 0xffffffffu : 0u, k1 = (rin && in1) ? 0xffffffffu : 0u;
                        st.x0 &= k0; st.x1 &= k0;
                        st.y0 &= k1; st.y1 &= k1;
                    }
                    st.t[0] = dpp_row_shl1(st.x0); st.t[1] = dpp_row_shl1(st.x1);       // the next pair's first column (d2)
                    st.t[2] = dpp_row_shl1(st.y0); st.t[3] = dpp_row_shl1(st.y1);       // ... and second column (d3)
                } else if constexpr (k == 5) {
                    st.q[0] = pk_sub_f16(st.x0, st.t[0]); st.q[1] = pk_sub_f16(st.x1, st.t[1]);      // V0 = d0 - d2
                    st.q[2] = pk_add_f16(st.y0, st.t[0]); st.q[3] = pk_add_f16(st.y1, st.t[1]);      // V1 = d1 + d2
                    st.q[4] = pk_sub_f16(st.t[0], st.y0); st.q[5] = pk_sub_f16(st.t[1], st.y1);      // V2 = d2 - d1
                    st.q[6] = pk_sub_f16(st.y0, st.t[2]); st.q[7] = pk_sub_f16(st.y1, st.t[3]);      // V3 = d1 - d3
                } else if constexpr (k == 6) {
                    // no lane exchange: every lane stores its own four channels of V0..V3 as 8-byte pieces (2-way bank conflicts --
                    // eight even pairs onto four units -- and still 0.7 % faster than four v_permlane16_swap and two 16-byte stores
                    // per row: -DTW_W128, profiles/r04_ab_results.txt block 9)
                    int pos = b10w + n;
                    pos = pos >= TW_BROWS ? pos - TW_BROWS : pos;
                    if (p < 15) {
                        char* const w0 = smem + w64 + (unsigned)pos * TW_BROWB;
#pragma unroll
                        for (int j = 0; j < 4; ++j) *(uint2*)(w0 + j * (TW_BROWB / 4)) = make_uint2(st.q[2 * j], st.q[2 * j + 1]);
                    }
                } else if constexpr (k == 7) {
                } else if constexpr (k == 8) {
                    const auto s02a = __builtin_amdgcn_permlane16_swap(st.q[0], st.q[4], false, false);
                    const auto s02b = __builtin_amdgcn_permlane16_swap(st.q[1], st.q[5], false, false);
                    const auto s13a = __builtin_amdgcn_permlane16_swap(st.q[2], st.q[6], false, false);
                    const auto s13b = __builtin_amdgcn_permlane16_swap(st.q[3], st.q[7], false, false);
                    st.q[0] = s02a[0]; st.q[1] = s02b[0]; st.q[2] = s02a[1]; st.q[3] = s02b[1];
                    st.q[4] = s13a[0]; st.q[5] = s13b[0]; st.q[6] = s13a[1]; st.q[7] = s13b[1];
                } else {
                    int pos = b10w + n;
                    pos = pos >= TW_BROWS ? pos - TW_BROWS : pos;
                    char* const w0 = smem + wlane + (unsigned)pos * wrow;
                    *(uint4*)w0 = make_uint4(st.q[0], st.q[1], st.q[2], st.q[3]);
                    *(uint4*)(w0 + wj) = make_uint4(st.q[4], st.q[5], st.q[6], st.q[7]);
                }
            };
            set_step(ey, b10);
            if (it < nsteps) {
                __builtin_amdgcn_s_setprio(TW_PRIO_KA);
                // steps that touch their plane's edge (uniform, few) mask what lies outside
                auto no_hook = [](auto) __attribute__((always_inline)) {};
                kloop(std::false_type{}, a6, [&](auto nc, auto kc) __attribute__((always_inline)) { slice(st2[0], std::false_type{}, nc, kc); }, no_hook, pre);
                __builtin_amdgcn_s_setprio(TW_PRIO_EA);
            }
            // the rows of step it + 1 (issued one iteration ago) are complete once only this phase's pieces are outstanding
            TW_STAMP(1);
            group_barrier();
            TW_STAMP(2);
            if (it < nsteps) {                     // the last row: beside the consumers' k-loop
                // two rows at a time, slice by slice: neighbouring instructions are independent of each other
                auto rest = [&](auto edge_tag) __attribute__((always_inline)) {
                    static_for<2>([&](auto rc) __attribute__((always_inline)) {
                        static_for<8>([&](auto kc) __attribute__((always_inline)) {
                            slice(st2[0], edge_tag, std::integral_constant<int, 2 * decltype(rc)::value>{}, kc);
                            slice(st2[1], edge_tag, std::integral_constant<int, 2 * decltype(rc)::value + 1>{}, kc);
                        });
                    });
                };
                if (edge) rest(std::true_type{}); else rest(std::false_type{});     // (uniform: most steps lie inside their plane)
            }
            e_own = load_a(it + 1 < nsteps ? it + 1 : nsteps - 1);
            TW_STAMP(3);
            a6 = a6 + 4 >= TW_AROWS ? a6 + 4 - TW_AROWS : a6 + 4;
            group_barrier();
            b10 = b10 + 4 >= TW_BROWS ? b10 + 4 - TW_BROWS : b10 + 4;
        }
    } else {
        // ---- group B: step s: k-loop in iteration s + 1 (phase Y), its rows 0..2 stored from inside that k-loop, row 3 in
        // phase X of iteration s + 2.  e_k = the entry of the step whose k-loop runs in this iteration, e_3 = of the one before.
        int a6 = 0, b10 = 0;
        uint4 e_k = make_uint4(0, 0, 0, 0), e_3 = make_uint4(0, 0, 0, 0);
        const int cg = lane >> 4, p = lane & 15;
        const int col = 2 * p + (cg & 1);
        // after the lane exchange lane (p, cg) holds 8 consecutive channels 16 wave + 8 (cg >> 1) .. of column col
        const unsigned olane = (unsigned)(col * PIXB + 32 * wave + 16 * (cg >> 1));
        char* const sink = (char*)a.sink + lane * PIXB;
        const unsigned flip0 = (s4[0] > 1.f ? 0x8000u : 0u) | (s4[1] > 1.f ? 0x80000000u : 0u);
        const unsigned flip1 = (s4[2] > 1.f ? 0x8000u : 0u) | (s4[3] > 1.f ? 0x80000000u : 0u);
        auto make_slice = [&](const uint4 e) __attribute__((always_inline)) {
            const unsigned ey = __builtin_amdgcn_readfirstlane(e.y), lo = __builtin_amdgcn_readfirstlane(e.x);
            const size_t off = ((unsigned long long)(ey & 0xffu) << 32) | lo;
            const int pitch = __builtin_amdgcn_readfirstlane(e.z) & 0xffffff;
            const int vy = (ey >> 8) & 7, vx = (ey >> 11) & 63, v0 = (ey >> 17) & 7;
            const bool fold = (ey >> 25) & 1u;                      // pairs 8..15 store into the second plane: 16 columns back, one plane on
            const unsigned fadd = fold ? (unsigned)__builtin_amdgcn_readfirstlane(e.w) : 0u;
            char* const obase = a.out_act + off + olane + ((fold && p >= 8) ? fadd : 0u);
            const bool colok = (fold ? (col & 15) : col) < vx;
            return [&, obase, pitch, vy, v0, colok](RowSt& st, auto nc, auto kc) __attribute__((always_inline)) {
                constexpr int n = decltype(nc)::value, k = decltype(kc)::value;
                if constexpr (k == 0) fin_sum(st, n, 0);
                else if constexpr (k == 1) fin_act(st, 0);
                else if constexpr (k == 2) fin_sum(st, n, 1);
                else if constexpr (k == 3) fin_act(st, 1);
                else if constexpr (k == 4) {
                    if constexpr (ACT == TW_ACT_F16_FLIP) {       // channels computed negated (slope > 1) get their sign back
                        st.x0 ^= flip0; st.y0 ^= flip0;
                        st.x1 ^= flip1; st.y1 ^= flip1;
                    }
                    const auto x = __builtin_amdgcn_permlane16_swap(st.x0, st.y0, false, false);
                    const auto y = __builtin_amdgcn_permlane16_swap(st.x1, st.y1, false, false);
                    st.q[0] = x[0]; st.q[1] = y[0]; st.q[2] = x[1]; st.q[3] = y[1];
                } else if constexpr (k == 5) {
                    char* dst = (n >= v0 && n < vy && colok) ? obase + (size_t)n * pitch : sink;
                    *(uint4*)dst = make_uint4(st.q[0], st.q[1], st.q[2], st.q[3]);
                }
            };
        };
        // the A half of step it + 2's entry (its input rows), fetched one iteration ahead like the producers' own entries
        uint4 e_dma = load_a(2 <= nsteps + TW_PAD_STEPS - 1 ? 2 : nsteps + TW_PAD_STEPS - 1);
        unsigned st_prev = 0;          // the previous iteration's epilogue ran (four stores behind that iteration's pieces)
        for (int it = 0; it < niter; ++it) {
            TW_STAMP(0);
            // Raw rows of step it + 2 -> slot it & 1, whose rows this group read (into registers) in the k-loop in front of barrier 2.
            // They are read again in iteration it + 1's k-loop; every wave waits for ITS pieces in front of that iteration's barrier 1.
            issue_rows(e_dma, it & 1, std::integral_constant<int, 0>{}, std::integral_constant<int, TW_DMA_PIECES>{});
            e_dma = load_a(it + 3 <= nsteps + TW_PAD_STEPS - 1 ? it + 3 : nsteps + TW_PAD_STEPS - 1);
            const unsigned st_cur = (unsigned)__builtin_amdgcn_readfirstlane((int)(it >= 2 ? (__builtin_amdgcn_readfirstlane(e_3.y) >> 24) & 1u : 0u));
            if (it >= 2 && ((__builtin_amdgcn_readfirstlane(e_3.y) >> 24) & 1u)) {      // row 3 of step it - 2
                auto sl3 = make_slice(e_3);
                static_for<2>([&](auto rc) __attribute__((always_inline)) {      // in pairs, slice by slice
                    static_for<6>([&](auto kc) __attribute__((always_inline)) {
                        sl3(st2[0], std::integral_constant<int, 2 * decltype(rc)::value>{}, kc);
                        sl3(st2[1], std::integral_constant<int, 2 * decltype(rc)::value + 1>{}, kc);
                    });
                });
            }
            e_k = load_b(it >= 1 ? it - 1 : 0);
            const unsigned kact = (it >= 1 && it <= nsteps) ? (__builtin_amdgcn_readfirstlane(e_k.y) >> 24) & 1u : 0u;
            int bp = b10 - 4 - 2;                  // block it - 1 begins at (4 (it - 1)) mod 10; the window starts two rows above
            bp = bp < 0 ? bp + TW_BROWS : bp;
            half8 pre[TW_PFF_B];
            TW_STAMP(1);
            // The pieces of iteration it - 1 (the rows of step it + 1, read behind this barrier) have landed once nothing OLDER than what
            // this wave has issued since is outstanding -- vmcnt counts a wave's loads and stores in order (tools/vmorder_bench.hip):
            // the previous iteration's stores (4, if its epilogue ran), this iteration's pieces (four; wave 0 has a fifth) and stores.
            // The count is by hand: it holds as long as an epilogue is EXACTLY four VMEM stores per wave and nothing else of this wave
            // goes through the vector memory path.  The instrumented build's stamps are global stores: it waits for
            // everything instead.
#ifdef UVA_INSTRUMENT
            asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
            st_prev = st_cur;
#else
            {
                constexpr int P0 = TW_DMA_PIECES, P = TW_DMA_PIECES - 1;
                const unsigned nst = st_prev + st_cur;
                if (wave == 0) {
                    if (nst == 2) asm volatile("s_waitcnt vmcnt(%0)" ::"n"(P0 + 8) : "memory");
                    else if (nst == 1) asm volatile("s_waitcnt vmcnt(%0)" ::"n"(P0 + 4) : "memory");
                    else asm volatile("s_waitcnt vmcnt(%0)" ::"n"(P0) : "memory");
                } else {
                    if (nst == 2) asm volatile("s_waitcnt vmcnt(%0)" ::"n"(P + 8) : "memory");
                    else if (nst == 1) asm volatile("s_waitcnt vmcnt(%0)" ::"n"(P + 4) : "memory");
                    else asm volatile("s_waitcnt vmcnt(%0)" ::"n"(P) : "memory");
                }
                st_prev = st_cur;
            }
#endif
            prefetch(std::true_type{}, bp, pre);   // window row 0 = the third row of block it - 2: written three phases ago
            asm volatile("s_barrier" ::: "memory");        // (kact above has waited for the step entry; nothing of this phase went to LDS)
            TW_STAMP(2);
            // The raw rows of step it + 1 (landed: the producers waited for them in front of barrier 1) -> A-ring.  FIRST: the
            // producers' epilogue runs beside it at full speed (beside a k-loop it gets one instruction through per MFMA), and
            // nothing is left behind the k-loop, when the LDS store path would be all that moves.
            [[maybe_unused]] auto raw_rows = [&]() __attribute__((always_inline)) {
                if (it + 1 < nsteps) {
                    int pos0 = a6 + 4 + 2;         // step it + 1's new rows follow its two shared ones
                    pos0 = pos0 >= TW_AROWS ? pos0 - TW_AROWS : pos0;
                    transform_rows((it + 1) & 1, pos0);
                }
            };
            unsigned solo = (unsigned)__builtin_amdgcn_readfirstlane((int)((kact ^ 1u) & (it + 1 < nsteps ? 1u : 0u)));
            asm volatile("" : "+s"(solo));         // (opaque: seen as the k-loop's `else`, hipcc lays the two out as one region and spills 115 registers)
            if (solo) {                            // (no k-loop to hide it in: the first iteration, a segment's fill step) -- half a
                int pos = a6 + 4 + 2 + wave;       // row at a time: the 32 registers of transform_row() do not fit beside the hook's
                pos = pos >= 2 * TW_AROWS ? pos - 2 * TW_AROWS : pos >= TW_AROWS ? pos - TW_AROWS : pos;
                const char* const rr = smem + TW_RAW + ((it + 1) & 1) * TW_RAWSLOTB + wave * TW_RAWROWB;
                transform_half(rr, pos, 0);
                __builtin_amdgcn_sched_barrier(0);
                transform_half(rr, pos, 1);
            }
            TW_STAMP(4);
            if (kact) {
                __builtin_amdgcn_s_setprio(TW_PRIO_KB);
                auto slk = make_slice(e_k);
                // The next step's raw row `wave` -> A-ring, cut into pieces that ride in this k-loop's instruction stream: four
                // ds_read_b128 (d0..d3 of one channel half) at fragment F0, then one V row (four v_pk_add_f16, one ds_write_b128) every
                // STRIDE fragments from F0 + GAP on -- an LDS round trip later --, then the other half.  Past the frame's last step
                // it transforms whatever the look-ahead DMA brought into rows nobody reads.
                int rpos = a6 + 4 + 2 + wave;
                rpos = rpos >= 2 * TW_AROWS ? rpos - 2 * TW_AROWS : rpos >= TW_AROWS ? rpos - TW_AROWS : rpos;
                const char* const rrow = smem + TW_RAW + ((it + 1) & 1) * TW_RAWSLOTB + wave * TW_RAWROWB;
                char* const vrow = smem + TW_ARING + rpos * TW_AROWB + vlane_c;
                half8 rd[4];
                auto raw_hook = [&](auto fc) __attribute__((always_inline)) {
                    constexpr int f = decltype(fc)::value;
                    constexpr int H1 = TW_RAW_F0 + TW_RAW_GAP + 3 * TW_RAW_STRIDE + 1;          // the second half's reads
                    static_assert(H1 + TW_RAW_GAP + 3 * TW_RAW_STRIDE < 48, "the transform ends inside the k-loop");
                    constexpr int t = f >= H1 ? 1 : 0, g = f - (t ? H1 : TW_RAW_F0);
                    const unsigned a_lo = t_lo ^ (t ? 64u : 0u), a_hi = t_hi ^ (t ? 64u : 0u);
                    if constexpr (g == 0) {
                        rd[0] = *(const half8*)(rrow + a_lo); rd[1] = *(const half8*)(rrow + a_lo + 17 * PIXB);
                        rd[2] = *(const half8*)(rrow + a_hi); rd[3] = *(const half8*)(rrow + a_hi + 17 * PIXB);
                    } else if constexpr (g == TW_RAW_GAP) *(half8*)(vrow + 0 * 2048 + t * 1024) = pk_sub(rd[0], rd[2]);
                    else if constexpr (g == TW_RAW_GAP + TW_RAW_STRIDE) *(half8*)(vrow + 3 * 2048 + t * 1024) = pk_sub(rd[1], rd[3]);
                    else if constexpr (g == TW_RAW_GAP + 2 * TW_RAW_STRIDE) *(half8*)(vrow + 1 * 2048 + t * 1024) = rd[1] + rd[2];
                    else if constexpr (g == TW_RAW_GAP + 3 * TW_RAW_STRIDE) *(half8*)(vrow + 2 * 2048 + t * 1024) = pk_sub(rd[2], rd[1]);
                };
                kloop(std::true_type{}, bp, [&](auto nc, auto kc) __attribute__((always_inline)) { slk(st2[0], nc, kc); }, raw_hook, pre);
                __builtin_amdgcn_s_setprio(TW_PRIO_EB);
            }
            e_3 = e_k;
            TW_STAMP(3);
            group_barrier();
            a6 = a6 + 4 >= TW_AROWS ? a6 + 4 - TW_AROWS : a6 + 4;
            b10 = b10 + 4 >= TW_BROWS ? b10 + 4 - TW_BROWS : b10 + 4;
        }
    }
    // nothing of the dummy look-ahead rows may land after the workgroup's LDS is released
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
}

}  // namespace uva
