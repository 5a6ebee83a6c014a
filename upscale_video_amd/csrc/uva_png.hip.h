// uva_png.hip.h -- PNG encoding of the result frame ON the MI355X: the `imwrite` side of the reference's per-frame hop
// (cv2.imwrite(frame.png), upscale/upscale_processing.py:288 and :519).
//
// The reference's pipeline is file to file, and after the net itself the PNG encoder is its largest per-frame cost: zlib
// at cv2's own settings needs ~170 ms of one core for a 3840x2160 frame, 70 times the 2.4 ms the net takes here.  The
// frame is already in HBM, the work is byte-wise and embarrassingly parallel, so it is done there:
//
//   * every block of R rows (R*(3w+1) <= 48 KiB) is one workgroup and one deflate block of its own;
//   * scanlines are filtered with Sub (what cv2 uses by default for 8-bit RGB at IMWRITE_PNG_STRATEGY_RLE), BGR -> RGB;
//   * the filtered bytes are Huffman-coded as literals with one of four FIXED code tables (Laplacian residual models of
//     different widths, built once on the host and written into each block as a dynamic-Huffman header); the workgroup
//     costs all four and takes the cheapest.  No LZ77 matches: on filtered video frames nearly all of zlib's gain at
//     level 1 / Z_RLE comes from the entropy code;
//   * every block ends with an empty stored block, which byte-aligns it (zlib's Z_SYNC_FLUSH): blocks are concatenated
//     by plain byte copies, the last one carries BFINAL;
//   * the workgroup also returns the Adler-32 of its stretch of filtered bytes; the host combines them.
//
//   * and the CRC-32 of its compressed bytes (the PNG chunk CRC is the framing's one pass over the data otherwise).
//
// Two kernels on the net's stream: png_deflate_kernel leaves every block in a slot of its own in HBM, png_pack_kernel
// concatenates them (HBM to HBM); the copy engine then takes the packed bytes to the caller's page-locked workspace on
// the download stream -- only the bytes produced cross PCIe, about half of the raw frame, while the next frame's net
// already runs and without a kernel waiting for PCIe.  The host adds the zlib / PNG framing and combines the checksums (uva_png_assemble: any thread,
// no GPU call, no pass over the data besides the one copy).  The stream is plain RFC 1950/1951: every PNG reader
// decodes it to the same pixels cv2.imwrite's file gives.
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>

#include <algorithm>
#include <cmath>
#include <cstring>
#include <queue>
#include <string>
#include <vector>

namespace uva {

constexpr int PNG_TABLES = 4;
constexpr int PNG_FILT_CAP = 48 * 1024;                 // filtered bytes per block
constexpr int PNG_HDR_CAP = 256;                        // bytes reserved for a block's dynamic-Huffman header
constexpr int PNG_STAGE_BYTES = PNG_FILT_CAP * 9 / 8 + PNG_HDR_CAP + 64;   // the flattest table costs <= 9 bits per byte
constexpr int PNG_SLOT_BYTES = (PNG_STAGE_BYTES + 255) / 256 * 256;
constexpr int PNG_META_WORDS = 8;                       // per block: compressed bytes, Adler s1, Adler s2, filtered bytes, CRC-32 of
                                                        // the compressed bytes, 3 spare
constexpr uint32_t PNG_ADLER_BASE = 65521;

constexpr int PNG_CRC_PIECE = 64;                       // bytes of a block one thread of the CRC phase takes
constexpr int PNG_CRC_LEVELS = 10;                      // 1024 pieces are combined pairwise in 10 levels

struct PngTables {
    uint32_t code[PNG_TABLES][260];                     // [symbol 0..256]: (bit-reversed code << 5) | length
    uint8_t hdr[PNG_TABLES][PNG_HDR_CAP];               // block header bits (BFINAL = 0, BTYPE = 2, the code lengths), LSB first
    int hdr_bits[PNG_TABLES];
    // CRC-32 combination: [level k][nibble j][value v] = (v << 4j) * x^(8 * PNG_CRC_PIECE * 2^k) mod P, so that appending
    // 2^k pieces to a partial CRC costs 8 lookups instead of a 32-step shift-and-add
    uint32_t crcmul[PNG_CRC_LEVELS][8][16];
};

// ---- host: code construction ------------------------------------------------------------------------------------
namespace png_detail {

// Huffman code lengths of at most maxlen bits: plain Huffman, frequencies halved (floor 1) until it fits
inline std::vector<int> huff_lengths(std::vector<uint64_t> freq, int maxlen)
{
    const int n = (int)freq.size();
    std::vector<int> len(n, 0);
    for (;;) {
        struct Node { uint64_t f; int l, r; };
        std::vector<Node> nodes;
        typedef std::pair<uint64_t, int> QE;
        std::priority_queue<QE, std::vector<QE>, std::greater<QE>> q;
        for (int i = 0; i < n; ++i)
            if (freq[i]) { nodes.push_back({freq[i], -1 - i, -1 - i}); q.push({freq[i], (int)nodes.size() - 1}); }
        if (nodes.size() == 1) { len[-1 - nodes[0].l] = 1; return len; }
        while (q.size() > 1) {
            const QE a = q.top(); q.pop();
            const QE b = q.top(); q.pop();
            nodes.push_back({a.first + b.first, a.second, b.second});
            q.push({a.first + b.first, (int)nodes.size() - 1});
        }
        int worst = 0;
        std::vector<std::pair<int, int>> st{{q.top().second, 0}};
        while (!st.empty()) {
            const std::pair<int, int> t = st.back(); st.pop_back();
            const Node& nd = nodes[t.first];
            if (nd.l < 0 && nd.l == nd.r) { len[-1 - nd.l] = t.second; worst = std::max(worst, t.second); continue; }
            st.push_back({nd.l, t.second + 1});
            st.push_back({nd.r, t.second + 1});
        }
        if (worst <= maxlen) return len;
        for (auto& f : freq)
            if (f) f = std::max<uint64_t>(1, f / 2);
    }
}

// RFC 1951 3.2.2: canonical codes from lengths; returned bit-reversed (deflate packs Huffman codes MSB first into an
// LSB-first bit stream)
inline std::vector<uint32_t> canonical_reversed(const std::vector<int>& len)
{
    int bl_count[16] = {0};
    for (int l : len) bl_count[l]++;
    bl_count[0] = 0;
    uint32_t next[16] = {0}, c = 0;
    for (int b = 1; b < 16; ++b) { c = (c + bl_count[b - 1]) << 1; next[b] = c; }
    std::vector<uint32_t> out(len.size(), 0);
    for (size_t i = 0; i < len.size(); ++i) {
        if (!len[i]) continue;
        uint32_t v = next[len[i]]++, r = 0;
        for (int b = 0; b < len[i]; ++b) r |= ((v >> b) & 1u) << (len[i] - 1 - b);
        out[i] = r;
    }
    return out;
}

struct BitWriter {
    std::vector<uint8_t> bytes;
    int nbits = 0;
    void put(uint32_t v, int n)
    {
        for (int i = 0; i < n; ++i, ++nbits) {
            if ((nbits & 7) == 0) bytes.push_back(0);
            bytes.back() |= (uint8_t)(((v >> i) & 1u) << (nbits & 7));
        }
    }
};

}  // namespace png_detail

// CRC-32 arithmetic in the reflected representation zlib uses (bit 31 = x^0): a * b mod P, and x^(8 * len) mod P.
// crc(A || B) = png_multmodp(x^(8 len B), crc(A)) ^ crc(B) for standard CRCs (zlib's crc32_combine).  a != 0.
__host__ __device__ inline uint32_t png_multmodp(uint32_t a, uint32_t b)
{
    uint32_t m = 1u << 31, p = 0;
    for (;;) {
        if (a & m) {
            p ^= b;
            if ((a & (m - 1)) == 0) break;
        }
        m >>= 1;
        b = (b & 1) ? (b >> 1) ^ 0xEDB88320u : b >> 1;
    }
    return p;
}
__host__ __device__ inline uint32_t png_x8n(uint64_t len)
{
    uint32_t sq = 0x00800000u;               // x^8
    uint32_t p = 1u << 31;                    // x^0
    while (len) {
        if (len & 1) p = png_multmodp(sq, p);
        sq = png_multmodp(sq, sq);
        len >>= 1;
    }
    return p;
}
__host__ __device__ inline uint32_t png_crc_combine(uint32_t crc1, uint32_t crc2, uint64_t len2) { return png_multmodp(png_x8n(len2), crc1) ^ crc2; }

// The four literal codes and their block headers.  Model: filtered byte v is the residual r = (int8)v with
// P(r) ~ exp(-|r| / sigma) (+ a floor so that every byte value has a code); sigma = 1, 3, 8 and "flat".
inline const PngTables& png_tables()
{
    static const PngTables T = [] {
        PngTables t;
        std::memset(&t, 0, sizeof t);
        const double sigmas[PNG_TABLES] = {1.0, 3.0, 8.0, 0.0};
        for (int k = 0; k < PNG_TABLES; ++k) {
            std::vector<uint64_t> f(257);
            for (int v = 0; v < 256; ++v) {
                const int m = std::min(v, 256 - v);
                f[v] = sigmas[k] > 0 ? (uint64_t)std::llround(1e7 * std::exp(-m / sigmas[k])) + 200 : 1000;
            }
            f[256] = 1;                                                  // end of block: once per 48 KiB
            const std::vector<int> len = png_detail::huff_lengths(f, k == PNG_TABLES - 1 ? 9 : 15);
            const std::vector<uint32_t> code = png_detail::canonical_reversed(len);
            for (int v = 0; v < 257; ++v) t.code[k][v] = (code[v] << 5) | (uint32_t)len[v];
            // header: BFINAL 0, BTYPE 10, HLIT = 0 (257 codes), HDIST = 0 (one distance code, of length 0: literals only),
            // the code lengths themselves coded one symbol each (no repeat codes) with their own Huffman code
            std::vector<int> seq(len.begin(), len.end());
            seq.push_back(0);
            std::vector<uint64_t> cf(19, 0);
            for (int l : seq) cf[l]++;
            const std::vector<int> clen = png_detail::huff_lengths(cf, 7);
            const std::vector<uint32_t> ccode = png_detail::canonical_reversed(clen);
            static const int order[19] = {16, 17, 18, 0, 8, 7, 9, 6, 10, 5, 11, 4, 12, 3, 13, 2, 14, 1, 15};
            int hclen = 19;
            while (hclen > 4 && clen[order[hclen - 1]] == 0) --hclen;
            png_detail::BitWriter w;
            w.put(0, 1); w.put(2, 2); w.put(0, 5); w.put(0, 5); w.put((uint32_t)(hclen - 4), 4);
            for (int i = 0; i < hclen; ++i) w.put((uint32_t)clen[order[i]], 3);
            for (int l : seq) w.put(ccode[l], clen[l]);
            if ((int)w.bytes.size() > PNG_HDR_CAP) std::abort();
            std::memcpy(t.hdr[k], w.bytes.data(), w.bytes.size());
            t.hdr_bits[k] = w.nbits;
        }
        for (int k = 0; k < PNG_CRC_LEVELS; ++k) {
            const uint32_t op = png_x8n((uint64_t)PNG_CRC_PIECE << k);
            for (int j = 0; j < 8; ++j)
                for (uint32_t v = 0; v < 16; ++v) t.crcmul[k][j][v] = png_multmodp(op, v << (4 * j));
        }
        return t;
    }();
    return T;
}

// Rows of one deflate block: its filtered bytes (3w + 1 per row) fit PNG_FILT_CAP, and its RAW rows -- staged in the
// `stage` area at a pitch of (3w + 3) & ~3 bytes before they are filtered -- fit PNG_STAGE_BYTES (for very narrow frames
// the pitch's padding makes that the tighter bound: w = 3, 4 900 rows would overrun the stage area into the tables).
inline int png_rows_per_block(int w)
{
    const int rawp = (3 * w + 3) & ~3;
    return std::max(1, std::min(PNG_FILT_CAP / (3 * w + 1), PNG_STAGE_BYTES / rawp));
}
inline int png_num_blocks(int h, int w) { const int r = png_rows_per_block(w); return (h + r - 1) / r; }
// bytes of the workspace a frame needs, in HBM ([meta: nblocks x 32 B, padded to 4 KiB][nblocks slots]) and in page-locked
// host memory ([the same meta][the blocks' bytes, concatenated]: the same bound)
inline size_t png_meta_bytes(int h, int w) { return ((size_t)png_num_blocks(h, w) * PNG_META_WORDS * 4 + 4095) / 4096 * 4096; }
inline size_t png_workspace_bytes(int h, int w) { return png_meta_bytes(h, w) + (size_t)png_num_blocks(h, w) * PNG_SLOT_BYTES; }

// ---- device ------------------------------------------------------------------------------------------------------
struct PngArgs {
    const uint8_t* src;           // u8 HWC BGR frame in HBM
    size_t stride;
    int h, w, rows_per_block, nblocks;
    const uint32_t* code;         // [PNG_TABLES][260]
    const uint8_t* hdr;           // [PNG_TABLES][PNG_HDR_CAP]
    const uint32_t* crcmul;       // [PNG_CRC_LEVELS][8][16]
    int hdr_bits[PNG_TABLES];
    uint32_t* meta;               // [nblocks][PNG_META_WORDS]   (HBM)
    uint8_t* slots;               // [nblocks][PNG_SLOT_BYTES]   (HBM)
};

constexpr int PNG_THREADS = 1024;     // 16 waves per CU: the per-thread byte loops are chains of dependent LDS reads, more waves hide them
static_assert(PNG_THREADS * PNG_CRC_PIECE >= PNG_STAGE_BYTES && (1 << PNG_CRC_LEVELS) == PNG_THREADS, "CRC phase: one piece per thread");
constexpr int png_lds_bytes() { return PNG_FILT_CAP + PNG_STAGE_BYTES + 260 * 4 + 260 * 8 + (PNG_THREADS + 8) * 8; }
static_assert(png_lds_bytes() <= 160 * 1024, "PNG kernel LDS budget");

__device__ __forceinline__ void png_or_bits(uint32_t* stage, unsigned long long pos, unsigned long long v, int n)
{
    // n <= 32 bits of v at bit position pos
    const unsigned sh = (unsigned)(pos & 31);
    const unsigned long long x = v << sh;
    uint32_t* p = stage + (pos >> 5);
    atomicOr(p, (uint32_t)x);
    if (sh + n > 32) atomicOr(p + 1, (uint32_t)(x >> 32));
}

__device__ __forceinline__ unsigned long long png_wave_sum(unsigned long long v)
{
#pragma unroll
    for (int d = 32; d >= 1; d >>= 1) v += __shfl_xor(v, d);
    return v;
}

__global__ __launch_bounds__(PNG_THREADS) void png_deflate_kernel(PngArgs a)
{
    extern __shared__ __attribute__((aligned(16))) char png_smem[];
    uint8_t* const filt = (uint8_t*)png_smem;
    uint32_t* const stage = (uint32_t*)(png_smem + PNG_FILT_CAP);
    uint32_t* const tbl = (uint32_t*)(png_smem + PNG_FILT_CAP + PNG_STAGE_BYTES);
    unsigned long long* const lens = (unsigned long long*)(tbl + 260);             // [symbol]: the four tables' lengths, 16 bits each
    unsigned long long* const red = lens + 260;                                    // reductions and the scan
    const int tid = threadIdx.x, b = blockIdx.x;
    const int r0 = b * a.rows_per_block, nr = min(a.rows_per_block, a.h - r0);
    const int rowb = 3 * a.w + 1, n = nr * rowb;

    for (int i = tid; i < 257; i += PNG_THREADS) {
        unsigned long long l = 0;
#pragma unroll
        for (int t = 0; t < PNG_TABLES; ++t) l |= (unsigned long long)(a.code[t * 260 + i] & 31u) << (16 * t);
        lens[i] = l;
    }
    if (tid < 8) red[tid] = 0;
    // ---- the block's rows -> LDS (the staging area, not needed yet), with as many loads in flight as registers allow:
    // a pixel-by-pixel walk over HBM is one ~1 us round trip per byte and thread ----
    uint8_t* const raw = (uint8_t*)stage;
    const int rawb = 3 * a.w, rawp = (rawb + 3) & ~3;
    if ((((uintptr_t)a.src | a.stride) & 3) == 0) {
        const int row_dw = (rawb + 3) / 4;                                // (the tail dword of a row reads into the next row: inside the frame
        constexpr int NB = 3;                                             //  except behind the last row, where it is read byte by byte)
        for (int row = 0; row < nr; ++row) {
            const uint32_t* const sp = (const uint32_t*)(a.src + (size_t)(r0 + row) * a.stride);
            uint32_t* const dp = (uint32_t*)(raw + (size_t)row * rawp);
            for (int base = 0; base < row_dw; base += NB * PNG_THREADS) {
                uint32_t v[NB];
#pragma unroll
                for (int k = 0; k < NB; ++k) {
                    const int j = base + k * PNG_THREADS + tid;
                    const bool last_partial = (r0 + row == a.h - 1) && 4 * j + 4 > rawb;
                    v[k] = 0;
                    if (j < row_dw) {
                        if (!last_partial) v[k] = sp[j];
                        else for (int e = 0; 4 * j + e < rawb; ++e) v[k] |= (uint32_t)((const uint8_t*)sp)[4 * j + e] << (8 * e);
                    }
                }
#pragma unroll
                for (int k = 0; k < NB; ++k) {
                    const int j = base + k * PNG_THREADS + tid;
                    if (j < row_dw) dp[j] = v[k];
                }
            }
        }
    } else {
        for (int row = 0; row < nr; ++row)
            for (int k = tid; k < rawb; k += PNG_THREADS) raw[(size_t)row * rawp + k] = a.src[(size_t)(r0 + row) * a.stride + k];
    }
    __syncthreads();

    // ---- filter (Sub, BGR -> RGB), cost under each table, Adler-32 sums ----
    unsigned long long c01 = 0, c23 = 0, s2 = 0;                          // costs of tables 0 | 1 << 32 and 2 | 3 << 32
    uint32_t s1 = 0;
    for (int row = 0; row < nr; ++row) {
        unsigned long long cost4 = 0;                                     // (a thread's <= 48 bytes of a row, <= 15 bits each: 16-bit fields hold)
        const uint8_t* const sp = raw + (size_t)row * rawp;
        uint8_t* const fp = filt + row * rowb;
        for (int k = tid; k < rowb; k += PNG_THREADS) {
            unsigned v = 1;                                               // filter type 1
            if (k > 0) {
                const int x = (k - 1) / 3, c = (k - 1) - 3 * x;
                const unsigned cur = sp[3 * x + 2 - c], left = x > 0 ? sp[3 * x - 1 - c] : 0;
                v = (cur - left) & 0xffu;
            }
            fp[k] = (uint8_t)v;
            cost4 += lens[v];
            s1 += v;
            s2 += (unsigned long long)((uint32_t)(n - (row * rowb + k)) * v);
        }
        c01 += (cost4 & 0xffffull) | ((cost4 & 0xffff0000ull) << 16);
        c23 += ((cost4 >> 32) & 0xffffull) | ((cost4 >> 16) & 0xffff00000000ull);
    }
    {
        c01 = png_wave_sum(c01);
        c23 = png_wave_sum(c23);
        const unsigned long long w1 = png_wave_sum((unsigned long long)s1), w2 = png_wave_sum(s2);
        if ((tid & 63) == 0) {
            atomicAdd(&red[0], c01 & 0xffffffffull); atomicAdd(&red[1], c01 >> 32);
            atomicAdd(&red[2], c23 & 0xffffffffull); atomicAdd(&red[3], c23 >> 32);
            atomicAdd(&red[4], w1);
            atomicAdd(&red[5], w2);
        }
    }
    __syncthreads();
    for (int i = tid; i < PNG_STAGE_BYTES / 4; i += PNG_THREADS) stage[i] = 0;    // the raw rows have been used up
    __syncthreads();
    int best = 0;
#pragma unroll
    for (int t = 1; t < PNG_TABLES; ++t)
        if (red[t] + (unsigned long long)a.hdr_bits[t] < red[best] + (unsigned long long)a.hdr_bits[best]) best = t;
    for (int i = tid; i < 257; i += PNG_THREADS) tbl[i] = a.code[best * 260 + i];
    const int hb = a.hdr_bits[best];
    for (int i = tid; i < (hb + 7) / 8; i += PNG_THREADS) atomicOr(stage + (i >> 2), (uint32_t)a.hdr[best * PNG_HDR_CAP + i] << (8 * (i & 3)));
    __syncthreads();

    // ---- bit offsets: every thread owns a contiguous span of the filtered bytes ----
    const int span = (n + PNG_THREADS - 1) / PNG_THREADS, sb = min(n, tid * span), se = min(n, sb + span);
    unsigned long long bits = 0;
    for (int i = sb; i < se; ++i) bits += tbl[filt[i]] & 31u;
    unsigned long long* const scan = red + 8;
    scan[tid] = bits;
    __syncthreads();
    for (int d = 1; d < PNG_THREADS; d <<= 1) {
        const unsigned long long v = tid >= d ? scan[tid - d] : 0;
        __syncthreads();
        scan[tid] += v;
        __syncthreads();
    }
    unsigned long long pos = (unsigned long long)hb + scan[tid] - bits;
    const unsigned long long body_end = (unsigned long long)hb + scan[PNG_THREADS - 1];

    // ---- encode the span ----
    unsigned long long acc = 0;
    int nacc = 0;
    for (int i = sb; i < se; ++i) {
        const uint32_t e = tbl[filt[i]];
        acc |= (unsigned long long)(e >> 5) << nacc;
        nacc += (int)(e & 31u);
        if (nacc >= 32) {
            png_or_bits(stage, pos, acc & 0xffffffffull, 32);
            pos += 32; acc >>= 32; nacc -= 32;
        }
    }
    if (nacc) png_or_bits(stage, pos, acc, nacc);
    // end of block, then the empty stored block that byte-aligns the stream (BFINAL on the frame's last block)
    unsigned long long total_bytes = 0;
    {
        const uint32_t eob = tbl[256];
        const unsigned long long p1 = body_end + (eob & 31u), p2 = (p1 + 3 + 7) / 8 * 8;
        total_bytes = p2 / 8 + 4;
        if (tid == 0) {
            png_or_bits(stage, body_end, eob >> 5, (int)(eob & 31u));
            png_or_bits(stage, p1, b == a.nblocks - 1 ? 1u : 0u, 3);
            png_or_bits(stage, p2, 0xffff0000ull, 32);                    // LEN = 0, NLEN = 0xffff
        }
    }
    __syncthreads();

    // ---- out: this block's bytes to its slot ----
    uint32_t* const out = (uint32_t*)(a.slots + (size_t)b * PNG_SLOT_BYTES);
    const int nw = (int)((total_bytes + 3) / 4);
    for (int i = tid; i < nw; i += PNG_THREADS) out[i] = stage[i];

    // ---- CRC-32 of the block's bytes, so that the host only combines: every thread takes one piece of PNG_CRC_PIECE
    // bytes, counted from the END of the block (the first pieces are empty: CRC 0, the neutral element); partial CRCs
    // are then combined pairwise, the right operand of a level-k combination always exactly 2^k pieces long, so the
    // multipliers are constants and come as nibble tables ----
    uint32_t* const ctab = tbl;                                           // (everybody is done with tbl and filt: the barrier above)
    uint32_t* const cmul = (uint32_t*)filt;
    {
        uint32_t c = (uint32_t)tid;
#pragma unroll
        for (int k = 0; k < 8; ++k) c = (c & 1) ? 0xEDB88320u ^ (c >> 1) : c >> 1;
        if (tid < 256) ctab[tid] = c;
        for (int i = tid; i < PNG_CRC_LEVELS * 128; i += PNG_THREADS) cmul[i] = a.crcmul[i];
    }
    __syncthreads();
    const int nbytes = (int)total_bytes;
    const int c0 = max(0, nbytes - (PNG_THREADS - tid) * PNG_CRC_PIECE), c1 = max(0, nbytes - (PNG_THREADS - 1 - tid) * PNG_CRC_PIECE);
    const uint8_t* const sb8 = (const uint8_t*)stage;
    uint32_t crc = 0xffffffffu;
    for (int i = c0; i < c1; ++i) crc = ctab[(crc ^ sb8[i]) & 0xffu] ^ (crc >> 8);
    crc = ~crc;
    uint32_t* const cred = (uint32_t*)(red + 8);                          // the scan array is free again
    cred[tid] = crc;
    __syncthreads();
#pragma unroll 1
    for (int k = 0; k < PNG_CRC_LEVELS; ++k) {
        const int d = 1 << k;
        if ((tid & (2 * d - 1)) == 0) {
            const uint32_t c = cred[tid];
            const uint32_t* const T = cmul + k * 128;
            uint32_t m = 0;
#pragma unroll
            for (int j = 0; j < 8; ++j) m ^= T[j * 16 + ((c >> (4 * j)) & 15u)];
            cred[tid] = m ^ cred[tid + d];
        }
        __syncthreads();
    }
    if (tid == 0) {
        uint32_t* const m = a.meta + (size_t)b * PNG_META_WORDS;
        m[0] = (uint32_t)total_bytes;
        m[1] = (uint32_t)((1 + red[4]) % PNG_ADLER_BASE);                 // Adler-32 of the block on its own
        m[2] = (uint32_t)((red[5] + (unsigned long long)n) % PNG_ADLER_BASE);
        m[3] = (uint32_t)n;
        m[4] = cred[0];
        m[5] = m[6] = m[7] = 0;
    }
}

// The blocks, concatenated (device to device; the copy engine takes the result to the caller's page-locked workspace:
// [meta, as png_deflate_kernel left it][the bytes]).  One workgroup per block; destination offsets are unaligned, the
// middle of every block goes out as aligned dwords.
struct PngPackArgs {
    const uint32_t* meta;         // HBM
    const uint8_t* slots;         // HBM
    int nblocks;
    uint32_t* out_meta;           // nullptr, or where the meta rows are to be copied as well
    uint8_t* out_data;
};
constexpr int PNG_PACK_THREADS = 256;

__global__ __launch_bounds__(PNG_PACK_THREADS) void png_pack_kernel(PngPackArgs a)
{
    __shared__ unsigned long long off_s;
    const int tid = threadIdx.x, b = blockIdx.x;
    if (tid == 0) off_s = 0;
    __syncthreads();
    unsigned long long part = 0;
    for (int j = tid; j < b; j += PNG_PACK_THREADS) part += a.meta[(size_t)j * PNG_META_WORDS];
    part = png_wave_sum(part);
    if ((tid & 63) == 0 && part) atomicAdd(&off_s, part);
    __syncthreads();
    const size_t off = (size_t)off_s;
    const uint32_t* const m = a.meta + (size_t)b * PNG_META_WORDS;
    const int nbytes = (int)m[0];
    if (a.out_meta && tid < PNG_META_WORDS) a.out_meta[(size_t)b * PNG_META_WORDS + tid] = m[tid];
    const uint8_t* const src = a.slots + (size_t)b * PNG_SLOT_BYTES;      // 256-byte aligned
    uint8_t* const dst = a.out_data + off;
    const int head = min(nbytes, (int)((4 - ((uintptr_t)dst & 3)) & 3));  // bytes until dst is dword aligned
    const int ndw = (nbytes - head) / 4, tail = nbytes - head - 4 * ndw;
    if (tid < head) dst[tid] = src[tid];
    if (tid < tail) dst[head + 4 * ndw + tid] = src[head + 4 * ndw + tid];
    const uint32_t* const s32 = (const uint32_t*)src;
    uint32_t* const d32 = (uint32_t*)(dst + head);
    for (int i = tid; i < ndw; i += PNG_PACK_THREADS) {
        // destination dword i = source bytes [head + 4i, head + 4i + 4)
        const uint32_t lo = s32[i], hi = head ? s32[i + 1] : 0u;          // (i + 1 stays inside the slot: PNG_SLOT_BYTES > PNG_STAGE_BYTES)
        d32[i] = head ? __builtin_amdgcn_alignbyte(hi, lo, (uint32_t)head) : lo;
    }
}

// ---- host: framing -----------------------------------------------------------------------------------------------
namespace png_detail {

inline const uint32_t (*crc_tables())[256]
{
    static uint32_t T[8][256];
    static const bool init = [] {
        for (uint32_t i = 0; i < 256; ++i) {
            uint32_t c = i;
            for (int k = 0; k < 8; ++k) c = (c & 1) ? 0xEDB88320u ^ (c >> 1) : c >> 1;
            T[0][i] = c;
        }
        for (uint32_t i = 0; i < 256; ++i)
            for (int t = 1; t < 8; ++t) T[t][i] = (T[t - 1][i] >> 8) ^ T[0][T[t - 1][i] & 0xff];
        return true;
    }();
    (void)init;
    return T;
}

// CRC-32 (ISO 3309, the PNG chunk CRC), slicing by 8; crc is the running value (start 0)
inline uint32_t crc32(uint32_t crc, const uint8_t* p, size_t n)
{
    const uint32_t (*T)[256] = crc_tables();
    uint32_t c = ~crc;
    while (n && ((uintptr_t)p & 7)) { c = T[0][(c ^ *p++) & 0xff] ^ (c >> 8); --n; }
    while (n >= 8) {
        uint32_t a, b;
        std::memcpy(&a, p, 4);
        std::memcpy(&b, p + 4, 4);
        a ^= c;
        c = T[7][a & 0xff] ^ T[6][(a >> 8) & 0xff] ^ T[5][(a >> 16) & 0xff] ^ T[4][a >> 24] ^
            T[3][b & 0xff] ^ T[2][(b >> 8) & 0xff] ^ T[1][(b >> 16) & 0xff] ^ T[0][b >> 24];
        p += 8; n -= 8;
    }
    while (n--) c = T[0][(c ^ *p++) & 0xff] ^ (c >> 8);
    return ~c;
}

// zlib's adler32_combine: Adler-32 of A||B from Adler(A), Adler(B) and len(B)
inline uint32_t adler_combine(uint32_t a1, uint32_t a2, uint64_t len2)
{
    const uint32_t B = PNG_ADLER_BASE;
    const uint32_t rem = (uint32_t)(len2 % B);
    uint32_t s1 = a1 & 0xffff, s2 = (uint32_t)(((uint64_t)rem * s1) % B);
    s1 += (a2 & 0xffff) + B - 1;
    s2 += (a1 >> 16) + (a2 >> 16) + B - rem;
    if (s1 >= B) s1 -= B;
    if (s1 >= B) s1 -= B;
    if (s2 >= (B << 1)) s2 -= (B << 1);
    if (s2 >= B) s2 -= B;
    return s1 | (s2 << 16);
}

inline void be32(uint8_t* p, uint32_t v) { p[0] = (uint8_t)(v >> 24); p[1] = (uint8_t)(v >> 16); p[2] = (uint8_t)(v >> 8); p[3] = (uint8_t)v; }

}  // namespace png_detail

// The PNG file image of an h x w frame from the workspace the kernel filled.  Returns 0 and *len, or non-zero with
// `err` set (workspace inconsistent, out too small).
inline int png_assemble(const uint8_t* workspace, int h, int w, uint8_t* out, size_t cap, size_t* len, std::string& err)
{
    using namespace png_detail;
    const int nb = png_num_blocks(h, w);
    const uint32_t* meta = (const uint32_t*)workspace;
    const uint8_t* bytes = workspace + png_meta_bytes(h, w);
    size_t total = 0;
    uint64_t filtered = 0;
    for (int b = 0; b < nb; ++b) {
        const uint32_t* m = meta + (size_t)PNG_META_WORDS * b;
        if (m[0] == 0 || m[0] > (uint32_t)PNG_SLOT_BYTES) { err = "PNG workspace: block size out of range (kernel not run?)"; return 1; }
        total += m[0];
        filtered += m[3];
    }
    if (filtered != (uint64_t)h * (3 * (uint64_t)w + 1)) { err = "PNG workspace: block lengths do not add up to the frame"; return 1; }
    const size_t idat = 2 + total + 4, need = 8 + 25 + 12 + idat + 12;
    if (len) *len = need;
    if (!out || cap < need) { err = "PNG output buffer too small"; return 1; }
    uint8_t* p = out;
    static const uint8_t sig[8] = {0x89, 'P', 'N', 'G', '\r', '\n', 0x1a, '\n'};
    std::memcpy(p, sig, 8); p += 8;
    be32(p, 13); std::memcpy(p + 4, "IHDR", 4);
    be32(p + 8, (uint32_t)w); be32(p + 12, (uint32_t)h);
    p[16] = 8; p[17] = 2; p[18] = 0; p[19] = 0; p[20] = 0;               // 8-bit RGB, deflate, adaptive filtering, no interlace
    be32(p + 21, crc32(0, p + 4, 17)); p += 25;
    be32(p, (uint32_t)idat); std::memcpy(p + 4, "IDAT", 4);
    uint8_t* const data = p + 8;
    data[0] = 0x78; data[1] = 0x01;                                       // zlib: deflate, 32 KiB window, fastest
    std::memcpy(data + 2, bytes, total);
    uint8_t* q = data + 2 + total;
    uint32_t adler = 1;
    uint32_t crc = crc32(0, p + 4, 4 + 2);                                // "IDAT" + the zlib header; the blocks' CRCs come from the GPU
    for (int b = 0; b < nb; ++b) {
        const uint32_t* m = meta + (size_t)PNG_META_WORDS * b;
        adler = adler_combine(adler, m[1] | (m[2] << 16), m[3]);
        crc = png_crc_combine(crc, m[4], m[0]);
    }
    be32(q, adler);
    crc = png_crc_combine(crc, crc32(0, q, 4), 4);
    q += 4;
    be32(q, crc); q += 4;
    be32(q, 0); std::memcpy(q + 4, "IEND", 4); be32(q + 8, crc32(0, q + 4, 4));
    return 0;
}

// Host restatement of png_deflate_kernel + png_pack_kernel, block for block and bit for bit (test hook: the CPU suite checks tables, headers
// and framing with it against a PNG reader, the GPU suite checks the kernel against it).  Not a product path.
inline void png_deflate_host(const uint8_t* src, size_t stride, int h, int w, uint8_t* workspace)
{
    const PngTables& T = png_tables();
    const int R = png_rows_per_block(w), nb = png_num_blocks(h, w), rowb = 3 * w + 1;
    uint32_t* meta = (uint32_t*)workspace;
    uint8_t* bytes = workspace + png_meta_bytes(h, w);
    std::vector<uint8_t> filt;
    for (int b = 0; b < nb; ++b) {
        const int r0 = b * R, nr = std::min(R, h - r0), n = nr * rowb;
        filt.assign(n, 0);
        uint64_t cost[PNG_TABLES] = {0}, s1 = 0, s2 = 0;
        for (int row = 0; row < nr; ++row)
            for (int k = 0; k < rowb; ++k) {
                unsigned v = 1;
                if (k > 0) {
                    const int x = (k - 1) / 3, c = (k - 1) - 3 * x;
                    const uint8_t* sp = src + (size_t)(r0 + row) * stride;
                    v = ((unsigned)sp[3 * x + 2 - c] - (x > 0 ? (unsigned)sp[3 * x - 1 - c] : 0u)) & 0xffu;
                }
                filt[row * rowb + k] = (uint8_t)v;
                for (int t = 0; t < PNG_TABLES; ++t) cost[t] += T.code[t][v] & 31u;
                s1 += v;
                s2 += (uint64_t)(n - (row * rowb + k)) * v;
            }
        int best = 0;
        for (int t = 1; t < PNG_TABLES; ++t)
            if (cost[t] + (uint64_t)T.hdr_bits[t] < cost[best] + (uint64_t)T.hdr_bits[best]) best = t;
        png_detail::BitWriter bw;
        for (int i = 0; i < T.hdr_bits[best]; ++i) bw.put((T.hdr[best][i >> 3] >> (i & 7)) & 1u, 1);
        for (int i = 0; i < n; ++i) bw.put(T.code[best][filt[i]] >> 5, (int)(T.code[best][filt[i]] & 31u));
        bw.put(T.code[best][256] >> 5, (int)(T.code[best][256] & 31u));
        bw.put(b == nb - 1 ? 1u : 0u, 3);
        while (bw.nbits & 7) bw.put(0, 1);
        bw.put(0xffff0000u, 32);
        std::memcpy(bytes, bw.bytes.data(), bw.bytes.size());
        bytes += bw.bytes.size();
        uint32_t* m = meta + (size_t)PNG_META_WORDS * b;
        m[0] = (uint32_t)bw.bytes.size();
        m[1] = (uint32_t)((1 + s1) % PNG_ADLER_BASE);
        m[2] = (uint32_t)((s2 + (uint64_t)n) % PNG_ADLER_BASE);
        m[3] = (uint32_t)n;
        m[4] = png_detail::crc32(0, bw.bytes.data(), bw.bytes.size());
        m[5] = m[6] = m[7] = 0;
    }
}

}  // namespace uva
