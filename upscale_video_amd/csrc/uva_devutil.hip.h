// uva_devutil.hip.h -- small device-side helpers shared by every kernel file of libuva (gfx950 only): vector types,
// LDS-DMA, scalar-cache loads, barriers with explicit wait counts, packed fp16 arithmetic.  Split out of
// uva_kernels.hip.h so that a kernel can live in a translation unit of its own (uva_wino.hip).
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>

#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <mutex>
#include <type_traits>
#include <utility>

namespace uva {

// Host side.  The library's A/B and debug switches (kernel choice, schedule variants, several of
// which change the output bytes) are environment variables, and a drop-in library must not change its numerics because of
// a stray variable in a user's shell: they are honoured only under the explicit opt-in UVA_DEBUG_SWITCHES=1 (the tests
// and tools set it).  Without the opt-in a set switch is ignored and named once on stderr.
inline const char* debug_env(const char* name)
{
    const char* const v = std::getenv(name);
    if (!v) return nullptr;
    static const bool opt_in = [] { const char* e = std::getenv("UVA_DEBUG_SWITCHES"); return e && std::strcmp(e, "1") == 0; }();
    if (opt_in) return v;
    static std::mutex mu;
    std::lock_guard<std::mutex> lock(mu);
    static char warned[32][40];
    static int nwarned = 0;
    for (int i = 0; i < nwarned; ++i)
        if (std::strncmp(warned[i], name, sizeof warned[i] - 1) == 0) return nullptr;
    if (nwarned < 32) {
        std::strncpy(warned[nwarned], name, sizeof warned[0] - 1);
        ++nwarned;
        std::fprintf(stderr, "libuva: %s is set but IGNORED: debug switches need UVA_DEBUG_SWITCHES=1\n", name);
    }
    return nullptr;
}

// compile-time loop: f(std::integral_constant<int, 0>{}), ..., f(std::integral_constant<int, N-1>{}).
// Used where a loop body must see its index as a constant expression (if constexpr, builtin
// immediates) and where '#pragma unroll' would give up on the pre-unrolling size of the body.
template <typename F, int... I>
__device__ __forceinline__ void static_for_impl(F&& f, std::integer_sequence<int, I...>)
{
    (f(std::integral_constant<int, I>{}), ...);
}
template <int N, typename F>
__device__ __forceinline__ void static_for(F&& f)
{
    static_for_impl(static_cast<F&&>(f), std::make_integer_sequence<int, N>{});
}


typedef _Float16 half8 __attribute__((ext_vector_type(8)));
typedef _Float16 half4 __attribute__((ext_vector_type(4)));
typedef float f32x16 __attribute__((ext_vector_type(16)));
typedef float f32x4 __attribute__((ext_vector_type(4)));

typedef _Float16 half2v __attribute__((ext_vector_type(2)));
typedef float f32x2 __attribute__((ext_vector_type(2)));
typedef unsigned u32x4 __attribute__((ext_vector_type(4)));

// a - b on eight fp16 values as four v_pk_add_f16 with neg modifiers.  Written on the dwords: `a - b` on half8 is
// scalarised by hipcc (v_sub_f16 + v_sub_f16_sdwa + v_pack_b32_f16 per dword), and __builtin_bit_cast of a vector
// ELEMENT reads element 0 whatever the index -- hence the scalar temporaries.
__device__ __forceinline__ unsigned pk_sub_f16(unsigned a, unsigned b)
{
    return __builtin_bit_cast(unsigned, (half2v)(__builtin_bit_cast(half2v, a) - __builtin_bit_cast(half2v, b)));
}
__device__ __forceinline__ half8 pk_sub(half8 a, half8 b)
{
    const u32x4 x = __builtin_bit_cast(u32x4, a), y = __builtin_bit_cast(u32x4, b);
    u32x4 z;
#pragma unroll
    for (int i = 0; i < 4; ++i) {
        const unsigned xi = x[i], yi = y[i];
        z[i] = pk_sub_f16(xi, yi);
    }
    return __builtin_bit_cast(half8, z);
}

constexpr int PARAM_LDS = 768;   // bias[64] + slope[64] + PReLU med3 selector[64] floats

__device__ __forceinline__ unsigned lds_offset(const void* p)
{
    return (unsigned)(size_t)(const __attribute__((address_space(3))) char*)p;
}

// A wave-uniform pointer that lives in VGPRs (e.g. computed from LDS reads), moved to SGPRs.
__device__ __forceinline__ const char* uniform_ptr(const char* p)
{
    const unsigned long long v = (unsigned long long)p;
    const unsigned lo = __builtin_amdgcn_readfirstlane((unsigned)v);
    const unsigned hi = __builtin_amdgcn_readfirstlane((unsigned)(v >> 32));
    return (const char*)(((unsigned long long)hi << 32) | lo);
}

// An opaque copy of a lane-constant value: stops hipcc from hoisting everything derived from it out
// of the persistent tile loop (where it would be spilled to scratch and reloaded behind a vmcnt wait).
__device__ __forceinline__ int opaque(int v)
{
    asm volatile("" : "+v"(v));
    return v;
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

// Same, with the tile's base address in SGPRs and a 32-bit per-lane byte offset (the "saddr" form):
// no 64-bit vector address arithmetic per piece.
__device__ __forceinline__ void glds16_s(const void* sbase, unsigned voff, unsigned lds_dst)
{
    unsigned keep;
    asm volatile(
        "s_mov_b32 %0, m0\n\t"
        "s_mov_b32 m0, %3\n\t"
        "s_nop 0\n\t"
        "global_load_lds_dwordx4 %1, %2\n\t"
        "s_mov_b32 m0, %0"
        : "=&s"(keep)
        : "v"(voff), "s"(sbase), "s"(lds_dst));
}

// 16 bytes through the scalar cache: the address must be wave-uniform.  The constant address space
// makes hipcc emit s_load_dwordx4 (its own counter, lgkmcnt) instead of a vector load + readfirstlane.
__device__ __forceinline__ uint4 scalar_load16(const uint4* p)
{
    typedef unsigned u32x4 __attribute__((ext_vector_type(4)));
    typedef const u32x4 __attribute__((address_space(4))) * const_ptr;
    const u32x4 v = *(const_ptr)(unsigned long long)p;
    return make_uint4(v[0], v[1], v[2], v[3]);
}

// No v_pk_{add,mul,fma}_f32 in a kernel whose waves work beside another wave's MFMAs on the same SIMD: a packed-fp32 instruction gets
// NO issue slot while the other wave's v_mfma_f32_16x16x32_f16 run back to back, and 0.7 per MFMA beside a k-loop with LDS reads,
// where v_add_f32 / v_fma_f32 / v_pk_*_f16 / DPP moves get 2.3 (tools/coissue_bench.hip, profiles/r05_ab_results.txt block 13).
// Two scalar instructions in place of one packed one: the same roundings, the same bytes.
#ifndef UVA_NO_PK_F32         // (-DUVA_NO_PK_F32= : A/B builds with the packed instructions back)
#ifdef __HIP_DEVICE_COMPILE__ // (the host pass of hipcc does not know the feature and says so)
#define UVA_NO_PK_F32 __attribute__((target("no-packed-fp32-ops")))
#else
#define UVA_NO_PK_F32
#endif
#endif

__device__ __forceinline__ void group_barrier()
{
    asm volatile("s_waitcnt lgkmcnt(0)\n\ts_barrier" ::: "memory");
}

template <int KEEP>
__device__ __forceinline__ void dma_barrier()
{
    asm volatile("s_waitcnt vmcnt(%0) lgkmcnt(0)\n\ts_barrier" ::"n"(KEEP) : "memory");
}

struct Trunk2Step {                             // 32 bytes
    // A half: x = input halo origin byte offset (low 32), y = offset bits 32..39 | row mask << 8 (bit r:
    // intermediate row r of the block is inside the plane) | c_lo << 12 | c_hi << 18 (intermediate columns
    // [c_lo, c_hi) of the block are inside the plane) | active << 24, z = row pitch in bytes
    uint4 a;
    // B half: x = output origin byte offset (low 32), y = offset bits 32..39 | valid rows << 8 |
    // valid columns << 11 | active << 24, z = row pitch in bytes
    uint4 b;
};
static_assert(sizeof(Trunk2Step) == 32, "Trunk2Step layout");

}  // namespace uva
