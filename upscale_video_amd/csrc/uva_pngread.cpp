// uva_pngread.cpp -- the `imread` side of the reference's per-frame hop on the host (cv2.imread(frame.png),
// upscale/upscale_processing.py:263, :487), host only.
//
// With the result frames deflated on the GPU (uva_png.hip.h) the PNG route is bound by DECODING the inputs, and a PNG
// reader spends most of its time in zlib's inflate (one serial bit stream per file: nothing for a GPU).  This is a
// from-scratch inflate in the style of the fast table decoders (64-bit bit buffer refilled with one unaligned load,
// 11-bit first-level tables whose entries carry base value and extra-bit count, literals and matches decoded without
// per-symbol bounds checks while both buffers have slack, word-wise match copies), plus PNG un-filtering straight into
// cv2's BGR layout.  RFC 1950 / 1951 / PNG 1.2; every check a careful reader makes is made (chunk CRCs, Adler-32, stream
// length), a corrupt file is an error, never a crash.  8-bit RGB / RGBA / grey (+alpha), non-interlaced -- what ffmpeg's
// `%d.extract.png` and cv2.imwrite produce; anything else returns 2 and the caller uses a general reader.
#include <cstddef>
#include <cstdint>
#include <cstring>
#include <string>
#include <algorithm>
#include <vector>

namespace uva {
namespace {

inline uint32_t rd_be32(const uint8_t* p) { return ((uint32_t)p[0] << 24) | ((uint32_t)p[1] << 16) | ((uint32_t)p[2] << 8) | p[3]; }

uint32_t crc_table[8][256];
// The decode threads of a worker (frame_pool.py, the GIL released inside the call) all come here on their first frame:
// a function-local static is initialised exactly once and its stores are visible to every thread that passes it.
void crc_init()
{
    static const bool ready = [] {
        for (uint32_t i = 0; i < 256; ++i) {
            uint32_t c = i;
            for (int k = 0; k < 8; ++k) c = (c & 1) ? 0xEDB88320u ^ (c >> 1) : c >> 1;
            crc_table[0][i] = c;
        }
        for (uint32_t i = 0; i < 256; ++i)
            for (int t = 1; t < 8; ++t) crc_table[t][i] = (crc_table[t - 1][i] >> 8) ^ crc_table[0][crc_table[t - 1][i] & 0xff];
        return true;
    }();
    (void)ready;
}
uint32_t crc32(uint32_t crc, const uint8_t* p, size_t n)
{
    uint32_t c = ~crc;
    while (n && ((uintptr_t)p & 7)) { c = crc_table[0][(c ^ *p++) & 0xff] ^ (c >> 8); --n; }
    while (n >= 8) {
        uint32_t a, b;
        std::memcpy(&a, p, 4);
        std::memcpy(&b, p + 4, 4);
        a ^= c;
        c = crc_table[7][a & 0xff] ^ crc_table[6][(a >> 8) & 0xff] ^ crc_table[5][(a >> 16) & 0xff] ^ crc_table[4][a >> 24] ^
            crc_table[3][b & 0xff] ^ crc_table[2][(b >> 8) & 0xff] ^ crc_table[1][(b >> 16) & 0xff] ^ crc_table[0][b >> 24];
        p += 8; n -= 8;
    }
    while (n--) c = crc_table[0][(c ^ *p++) & 0xff] ^ (c >> 8);
    return ~c;
}

uint32_t adler32(const uint8_t* p, size_t n)
{
    // 32-byte blocks: a' = a + sum p[i], b' = b + 32 a + sum (32 - i) p[i]; the two sums have no carried dependency and
    // vectorise, the byte-by-byte form is a chain of two dependent additions per byte
    uint32_t a = 1, b = 0;
    while (n) {
        size_t k = n < 5536 ? n : 5536;     // a multiple of 32 below the largest run that cannot overflow 32 bits (5552)
        n -= k;
        while (k >= 32) {
            uint32_t s1 = 0, s2 = 0;
            for (int i = 0; i < 32; ++i) { s1 += p[i]; s2 += (uint32_t)(32 - i) * p[i]; }
            b += 32 * a + s2;
            a += s1;
            p += 32; k -= 32;
        }
        while (k--) { a += *p++; b += a; }
        a %= 65521; b %= 65521;
    }
    return a | (b << 16);
}

// ---- inflate ------------------------------------------------------------------------------------------------------
// Table entry: bits 0..3 code length consumed at this level (or, for a link, the sub-table's index width),
// bits 4..7 kind, bits 8..12 extra-bit count, bits 16..31 base value / sub-table offset.
enum { K_LITERAL = 1, K_LENGTH = 2, K_EOB = 3, K_LINK = 4, K_DIST = 5, K_INVALID = 0,
       K_LITERAL2 = 9 };     // two literals in one entry (value = first | second << 8): kind & 7 == K_LITERAL for both
constexpr int LIT_BITS = 11, DIST_BITS = 8;
inline uint32_t mk(int len, int kind, int extra, int value) { return (uint32_t)len | ((uint32_t)kind << 4) | ((uint32_t)extra << 8) | ((uint32_t)value << 16); }

const uint16_t len_base[29] = {3, 4, 5, 6, 7, 8, 9, 10, 11, 13, 15, 17, 19, 23, 27, 31, 35, 43, 51, 59, 67, 83, 99, 115, 131, 163, 195, 227, 258};
const uint8_t len_extra[29] = {0, 0, 0, 0, 0, 0, 0, 0, 1, 1, 1, 1, 2, 2, 2, 2, 3, 3, 3, 3, 4, 4, 4, 4, 5, 5, 5, 5, 0};
const uint16_t dist_base[30] = {1, 2, 3, 4, 5, 7, 9, 13, 17, 25, 33, 49, 65, 97, 129, 193, 257, 385, 513, 769, 1025, 1537, 2049, 3073, 4097, 6145, 8193, 12289, 16385, 24577};
const uint8_t dist_extra[30] = {0, 0, 0, 0, 1, 1, 2, 2, 3, 3, 4, 4, 5, 5, 6, 6, 7, 7, 8, 8, 9, 9, 10, 10, 11, 11, 12, 12, 13, 13};

struct Table {
    std::vector<uint32_t> e;
    int root = 0;
};

inline uint32_t rev(uint32_t v, int n)
{
    uint32_t r = 0;
    for (int i = 0; i < n; ++i) r |= ((v >> i) & 1u) << (n - 1 - i);
    return r;
}

// canonical Huffman decode table from code lengths; is_dist selects the payload encoding.  Returns false for an
// over-subscribed code, or an incomplete one other than the single-code cases deflate allows.
bool build_table(const uint8_t* lens, int n, int root, bool is_dist, Table& t, bool pair_literals = false)
{
    int count[16] = {0};
    for (int i = 0; i < n; ++i) count[lens[i]]++;
    count[0] = 0;
    int maxlen = 0;
    for (int l = 1; l < 16; ++l) if (count[l]) maxlen = l;
    t.root = root;
    t.e.assign((size_t)1 << root, mk(0, K_INVALID, 0, 0));
    if (maxlen == 0) return true;                     // no codes at all: fine as long as none is used
    long long left = 1;
    for (int l = 1; l < 16; ++l) {
        left <<= 1;
        left -= count[l];
        if (left < 0) return false;
    }
    if (left > 0 && !(count[1] == 1 && maxlen == 1)) {
        // incomplete: zlib accepts that only for a lone 1-bit code (a distance alphabet with one code)
        return false;
    }
    uint32_t next[16] = {0}, code = 0;
    for (int l = 1; l < 16; ++l) { code = (code + (uint32_t)count[l - 1]) << 1; next[l] = code; }
    auto payload = [&](int sym, int len) -> uint32_t {
        if (is_dist) return sym < 30 ? mk(len, K_DIST, dist_extra[sym], dist_base[sym]) : mk(len, K_INVALID, 0, 0);
        if (sym < 256) return mk(len, K_LITERAL, 0, sym);
        if (sym == 256) return mk(len, K_EOB, 0, 0);
        return sym < 286 ? mk(len, K_LENGTH, len_extra[sym - 257], len_base[sym - 257]) : mk(len, K_INVALID, 0, 0);
    };
    // sub-tables: one per distinct root-bit prefix of the long codes, sized for the longest code under it
    static thread_local std::vector<int> sub_bits;             // (scratch reused from block to block: a 1080p frame has hundreds)
    sub_bits.assign((size_t)1 << root, 0);
    {
        uint32_t nx[16];
        std::memcpy(nx, next, sizeof nx);
        for (int i = 0; i < n; ++i) {
            const int l = lens[i];
            if (l <= root) { if (l) nx[l]++; continue; }
            const uint32_t c = nx[l]++;
            const uint32_t prefix = rev(c >> (l - root), root);
            sub_bits[prefix] = std::max(sub_bits[prefix], l - root);
        }
    }
    static thread_local std::vector<uint32_t> sub_off;
    sub_off.assign((size_t)1 << root, 0);
    for (size_t p = 0; p < sub_bits.size(); ++p)
        if (sub_bits[p]) {
            sub_off[p] = (uint32_t)t.e.size();
            t.e[p] = mk(sub_bits[p], K_LINK, 0, 0) | ((uint32_t)(t.e.size() >> 0) << 16);
            if (t.e.size() >= 65536) return false;
            t.e.resize(t.e.size() + ((size_t)1 << sub_bits[p]), mk(0, K_INVALID, 0, 0));
        }
    for (int i = 0; i < n; ++i) {
        const int l = lens[i];
        if (!l) continue;
        const uint32_t c = next[l]++;
        if (l <= root) {
            const uint32_t r = rev(c, l);
            const uint32_t ent = payload(i, l);
            for (uint32_t k = r; k < ((uint32_t)1 << root); k += (uint32_t)1 << l) t.e[k] = ent;
        } else {
            const uint32_t prefix = rev(c >> (l - root), root);
            const int sb = sub_bits[prefix], sl = l - root;
            const uint32_t r = rev(c & (((uint32_t)1 << sl) - 1), sl);
            const uint32_t ent = payload(i, sl);
            for (uint32_t k = r; k < ((uint32_t)1 << sb); k += (uint32_t)1 << sl) t.e[sub_off[prefix] + k] = ent;
        }
    }
    if (pair_literals) {
        // Filtered image data is mostly literals with short codes, and a literal costs one dependent table look-up: where
        // the root index holds a literal AND the whole code of a second one, the entry delivers both.  From the top down:
        // the entry looked up for the second symbol (index i >> len < i) is then still the single one.
        for (uint32_t i = (uint32_t)1 << root; i-- > 0;) {
            const uint32_t e = t.e[i];
            if (((e >> 4) & 15) != K_LITERAL) continue;
            const int l1 = (int)(e & 15);
            const uint32_t e2 = t.e[i >> l1];
            if (((e2 >> 4) & 15) == K_LITERAL && l1 + (int)(e2 & 15) <= root)
                t.e[i] = mk(l1 + (int)(e2 & 15), K_LITERAL2, 0, (int)((e >> 16) | ((e2 >> 16) << 8)));
        }
    }
    return true;
}

struct Bits {
    const uint8_t* p;
    const uint8_t* end;
    uint64_t buf = 0;
    int n = 0;               // valid bits in buf
    // After refill at least 56 bits are valid while input remains; past the end zeros are shifted in and `over`
    // counts the bytes invented, so that a stream that really needs them is reported as truncated.
    size_t over = 0;
    inline void refill()
    {
        if (end - p >= 8) {
            uint64_t v;
            std::memcpy(&v, p, 8);
            buf |= v << n;
            const int take = (63 - n) >> 3;
            p += take;
            n += take * 8;
        } else {
            while (n <= 56) {
                if (p < end) buf |= (uint64_t)*p++ << n;
                else ++over;
                n += 8;
            }
        }
    }
    // true once bits that do not exist have been CONSUMED (reading ahead past the end is normal)
    inline bool ran_dry() const { return over && (size_t)(n < 0 ? 0 : n) < over * 8; }
    inline uint32_t peek(int k) const { return (uint32_t)(buf & (((uint64_t)1 << k) - 1)); }
    inline void drop(int k) { buf >>= k; n -= k; }
    inline uint32_t take(int k) { const uint32_t v = peek(k); drop(k); return v; }
};

// RFC 1951.  out must hold exactly out_len bytes: more or fewer decoded bytes are an error.
bool inflate_raw(const uint8_t* in, size_t in_len, uint8_t* out, size_t out_len, std::string& err)
{
    Bits b{in, in + in_len};
    uint8_t* o = out;
    uint8_t* const oend = out + out_len;
    Table lit, dist;
    static const uint8_t clorder[19] = {16, 17, 18, 0, 8, 7, 9, 6, 10, 5, 11, 4, 12, 3, 13, 2, 14, 1, 15};
    for (;;) {
        b.refill();
        const uint32_t final = b.take(1), type = b.take(2);
        if (type == 0) {
            b.drop(b.n & 7);
            b.refill();
            const uint32_t len = b.take(16), nlen = b.take(16);
            if ((len ^ 0xffffu) != nlen) { err = "inflate: stored block length check failed"; return false; }
            // bytes still in the bit buffer first, then straight from the input
            uint32_t left = len;
            if (b.ran_dry()) { err = "inflate: truncated stream"; return false; }
            while (left && b.n >= 8 && (size_t)b.n > b.over * 8) {
                if (o == oend) { err = "inflate: more data than the image holds"; return false; }
                *o++ = (uint8_t)b.take(8); --left;
            }
            if (b.ran_dry()) { err = "inflate: truncated stream"; return false; }
            if (left) {
                if (b.p > b.end || (size_t)(b.end - b.p) < left) { err = "inflate: truncated stream"; return false; }
                if ((size_t)(oend - o) < left) { err = "inflate: more data than the image holds"; return false; }
                std::memcpy(o, b.p, left);
                o += left; b.p += left;
                b.buf = 0; b.n = 0;          // (the buffer was drained to the byte; what it read ahead is re-read)
            }
        } else if (type == 1 || type == 2) {
            uint8_t lens[320];
            int nlit, ndist;
            if (type == 1) {
                for (int i = 0; i < 144; ++i) lens[i] = 8;
                for (int i = 144; i < 256; ++i) lens[i] = 9;
                for (int i = 256; i < 280; ++i) lens[i] = 7;
                for (int i = 280; i < 288; ++i) lens[i] = 8;
                nlit = 288; ndist = 30;
                for (int i = 0; i < 30; ++i) lens[288 + i] = 5;
                lens[288 + 30] = lens[288 + 31] = 5;
                if (!build_table(lens, 288, LIT_BITS, false, lit, true)) { err = "inflate: internal table error"; return false; }
                uint8_t dl[32];
                for (int i = 0; i < 32; ++i) dl[i] = 5;
                if (!build_table(dl, 32, DIST_BITS, true, dist)) { err = "inflate: internal table error"; return false; }
            } else {
                nlit = (int)b.take(5) + 257; ndist = (int)b.take(5) + 1;
                const int ncl = (int)b.take(4) + 4;
                if (nlit > 286 || ndist > 30) { err = "inflate: too many length or distance codes"; return false; }
                uint8_t cl[19] = {0};
                for (int i = 0; i < ncl; ++i) { b.refill(); cl[clorder[i]] = (uint8_t)b.take(3); }
                static thread_local Table ct;
                if (!build_table(cl, 19, 7, false, ct)) { err = "inflate: bad code-length code"; return false; }
                int i = 0;
                while (i < nlit + ndist) {
                    b.refill();
                    const uint32_t e = ct.e[b.peek(7)];
                    if (((e >> 4) & 15) == K_INVALID || (e & 15) == 0) { err = "inflate: bad code-length symbol"; return false; }
                    b.drop((int)(e & 15));
                    // the code-length table was built with the literal payload encoding: symbol = value field
                    const int sym = (int)(e >> 16);
                    if (sym < 16) lens[i++] = (uint8_t)sym;
                    else {
                        int rep, val = 0;
                        if (sym == 16) { if (i == 0) { err = "inflate: repeat with no previous length"; return false; } val = lens[i - 1]; rep = 3 + (int)b.take(2); }
                        else if (sym == 17) rep = 3 + (int)b.take(3);
                        else rep = 11 + (int)b.take(7);
                        if (i + rep > nlit + ndist) { err = "inflate: code lengths overrun"; return false; }
                        while (rep--) lens[i++] = (uint8_t)val;
                    }
                }
                if (lens[256] == 0) { err = "inflate: no end-of-block code"; return false; }
                if (!build_table(lens, nlit, LIT_BITS, false, lit, true)) { err = "inflate: bad literal/length code"; return false; }
                if (!build_table(lens + nlit, ndist, DIST_BITS, true, dist)) { err = "inflate: bad distance code"; return false; }
            }
            if (b.ran_dry()) { err = "inflate: truncated stream"; return false; }
            const uint32_t* const LT = lit.e.data();
            const uint32_t* const DT = dist.e.data();
            bool block_done = false;
            // Fast loop: while 16 input bytes and FAST_OUT output bytes remain, nothing in it needs a bounds check -- one
            // iteration emits at most 56 literals (one bit each from one refill) and one match of 258 bytes, whose
            // word-wise copy may overshoot by 7.  An iteration refills TWICE (top of the loop, and in front of the distance
            // code), each an 8-byte read after advancing by at most 7: p <= end - 16 at the top keeps both inside the
            // input and leaves p <= end on exit.  The bit buffer lives in registers.
            constexpr ptrdiff_t FAST_OUT = 56 + 258 + 8 + 8;
            if (b.end - b.p >= 16 && oend - o >= FAST_OUT) {
                uint64_t buf = b.buf;
                int nb = b.n;
                const uint8_t* p = b.p;
                const uint8_t* const in_safe = b.end - 16;
                uint8_t* const out_safe = oend - FAST_OUT;
                const uint32_t LIT = (uint32_t)K_LITERAL << 4, LITMASK = 0x70u;       // one literal or two
                while (p <= in_safe && o <= out_safe) {
                    uint64_t v;
                    std::memcpy(&v, p, 8);
                    buf |= v << nb;
                    p += (63 - nb) >> 3;
                    nb |= 56;
                    uint32_t e = LT[buf & ((1u << LIT_BITS) - 1)];
                    bool refill_first = false;
                    while ((e & LITMASK) == LIT) {
                        buf >>= (e & 15); nb -= (int)(e & 15);
                        o[0] = (uint8_t)(e >> 16);
                        o[1] = (uint8_t)(e >> 24);              // (a single literal's entry has a zero here: overwritten next)
                        o += 1 + ((e >> 7) & 1);
                        if (nb < 32) { refill_first = true; break; }
                        e = LT[buf & ((1u << LIT_BITS) - 1)];
                    }
                    if (refill_first) continue;
                    // here nb >= 32: a length code (<= 15 bits) and its extra bits (<= 5) fit
                    if ((e & 0xf0u) == ((uint32_t)K_LINK << 4)) {
                        buf >>= LIT_BITS; nb -= LIT_BITS;
                        e = LT[(e >> 16) + (uint32_t)(buf & ((1u << (e & 15)) - 1))];
                    }
                    const uint32_t kind = (e >> 4) & 15;
                    buf >>= (e & 15); nb -= (int)(e & 15);
                    if (kind == K_LITERAL) { *o++ = (uint8_t)(e >> 16); continue; }     // (from a sub-table: always single)
                    if (kind == K_EOB) { block_done = true; break; }
                    if (kind != K_LENGTH) { err = "inflate: invalid literal/length code"; return false; }
                    const uint32_t xl = (e >> 8) & 31;
                    const uint32_t len = (e >> 16) + (uint32_t)(buf & ((1u << xl) - 1));
                    buf >>= xl; nb -= (int)xl;
                    std::memcpy(&v, p, 8);                       // p <= in_safe + 7 = end - 9: inside the input
                    buf |= v << nb;
                    p += (63 - nb) >> 3;
                    nb |= 56;
                    uint32_t d = DT[buf & ((1u << DIST_BITS) - 1)];
                    if ((d & 0xf0u) == ((uint32_t)K_LINK << 4)) {
                        buf >>= DIST_BITS; nb -= DIST_BITS;
                        d = DT[(d >> 16) + (uint32_t)(buf & ((1u << (d & 15)) - 1))];
                    }
                    if (((d >> 4) & 15) != K_DIST) { err = "inflate: invalid distance code"; return false; }
                    buf >>= (d & 15); nb -= (int)(d & 15);
                    const uint32_t xd = (d >> 8) & 31;
                    const uint32_t distance = (d >> 16) + (uint32_t)(buf & ((1u << xd) - 1));
                    buf >>= xd; nb -= (int)xd;
                    if (distance > (size_t)(o - out)) { err = "inflate: distance reaches before the start of the data"; return false; }
                    const uint8_t* s = o - distance;
                    uint8_t* const stop = o + len;
                    if (distance >= 8) {
                        do { std::memcpy(&v, s, 8); std::memcpy(o, &v, 8); s += 8; o += 8; } while (o < stop);
                    } else if (distance == 1) {
                        std::memset(o, *s, len);
                    } else {
                        do { *o++ = *s++; } while (o < stop);
                    }
                    o = stop;
                }
                b.buf = buf; b.n = nb; b.p = p;
            }
            while (!block_done) {
                b.refill();
                if (__builtin_expect(b.over != 0, 0) && b.ran_dry()) { err = "inflate: truncated stream"; return false; }
                uint32_t e = LT[b.peek(LIT_BITS)];
                if (((e >> 4) & 15) == K_LINK) {
                    b.drop(LIT_BITS);
                    e = LT[(e >> 16) + b.peek((int)(e & 15))];
                }
                const int kind = (int)((e >> 4) & 15);
                b.drop((int)(e & 15));
                if (kind == K_LITERAL2) {
                    if (oend - o < 2) { err = "inflate: more data than the image holds"; return false; }
                    *o++ = (uint8_t)(e >> 16);
                    *o++ = (uint8_t)(e >> 24);
                    continue;
                }
                if (kind == K_LITERAL) {
                    if (o == oend) { err = "inflate: more data than the image holds"; return false; }
                    *o++ = (uint8_t)(e >> 16);
                    // two more literals from the same refill are the common case in image data
                    uint32_t e2 = LT[b.peek(LIT_BITS)];
                    if (((e2 >> 4) & 15) == K_LITERAL && o != oend) {
                        b.drop((int)(e2 & 15));
                        *o++ = (uint8_t)(e2 >> 16);
                        e2 = LT[b.peek(LIT_BITS)];
                        if (((e2 >> 4) & 15) == K_LITERAL && o != oend) {
                            b.drop((int)(e2 & 15));
                            *o++ = (uint8_t)(e2 >> 16);
                        }
                    }
                    continue;
                }
                if (kind == K_EOB) break;
                if (kind != K_LENGTH) { err = "inflate: invalid literal/length code"; return false; }
                const uint32_t len = (e >> 16) + b.take((int)((e >> 8) & 31));
                b.refill();
                uint32_t d = DT[b.peek(DIST_BITS)];
                if (((d >> 4) & 15) == K_LINK) {
                    b.drop(DIST_BITS);
                    d = DT[(d >> 16) + b.peek((int)(d & 15))];
                }
                if (((d >> 4) & 15) != K_DIST) { err = "inflate: invalid distance code"; return false; }
                b.drop((int)(d & 15));
                const uint32_t distance = (d >> 16) + b.take((int)((d >> 8) & 31));
                if (distance > (size_t)(o - out)) { err = "inflate: distance reaches before the start of the data"; return false; }
                if (len > (size_t)(oend - o)) { err = "inflate: more data than the image holds"; return false; }
                const uint8_t* s = o - distance;
                if (distance >= 8 && (size_t)(oend - o) >= len + 8) {
                    // word-wise: may write up to 7 bytes past the match, inside the buffer, overwritten by what follows
                    uint8_t* const stop = o + len;
                    do { uint64_t v; std::memcpy(&v, s, 8); std::memcpy(o, &v, 8); s += 8; o += 8; } while (o < stop);
                    o = stop;
                } else if (distance == 1) {
                    std::memset(o, *s, len);
                    o += len;
                } else {
                    for (uint32_t k = 0; k < len; ++k) o[k] = s[k];
                    o += len;
                }
            }
            if (b.ran_dry()) { err = "inflate: truncated stream"; return false; }
        } else {
            err = "inflate: reserved block type";
            return false;
        }
        if (final) break;
    }
    if (o != oend) { err = "inflate: less data than the image needs"; return false; }
    // the Adler-32 follows on the next byte boundary; hand the position back through `in_len`-relative arithmetic
    return true;
}

// position of the first byte after the deflate stream is not tracked above (the bit buffer reads ahead); zlib's trailer
// is therefore taken from the END of the IDAT payload, which is where it must be
bool zlib_decompress(const uint8_t* in, size_t n, uint8_t* out, size_t out_len, std::string& err)
{
    if (n < 6) { err = "zlib stream too short"; return false; }
    if ((in[0] & 15) != 8 || (in[0] >> 4) > 7 || ((in[0] << 8) | in[1]) % 31 || (in[1] & 32)) { err = "bad zlib header"; return false; }
    if (!inflate_raw(in + 2, n - 6, out, out_len, err)) return false;
    if (adler32(out, out_len) != rd_be32(in + n - 4)) { err = "zlib: Adler-32 mismatch"; return false; }
    return true;
}

inline int paeth(int a, int b, int c)
{
    const int p = a + b - c, pa = p > a ? p - a : a - p, pb = p > b ? p - b : b - p, pc = p > c ? p - c : c - p;
    return (pa <= pb && pa <= pc) ? a : (pb <= pc ? b : c);
}

// PNG filters undone in place (raw: h rows of 1 + w*bpp bytes), bpp = bytes per pixel
// bgr (bpp 3 only): the unfiltered row also goes out as B G R pixels, h rows of 3 w bytes -- in the same pass for Sub rows
// (what cv2.imwrite and this library's own encoder write), whose running pixel then lives in registers instead of being
// stored and reloaded three bytes later
bool unfilter(uint8_t* raw, int h, int w, int bpp, uint8_t* bgr, std::string& err)
{
    const size_t rowb = (size_t)w * bpp;
    std::vector<uint8_t> zero(rowb, 0);
    const uint8_t* prev = zero.data();
    for (int y = 0; y < h; ++y) {
        uint8_t* const row = raw + (size_t)y * (rowb + 1) + 1;
        uint8_t* const drow = bpp == 3 && bgr ? bgr + (size_t)y * rowb : nullptr;
        if (drow && row[-1] == 1) {
            unsigned r = 0, g = 0, b = 0;
            for (size_t i = 0; i < rowb; i += 3) {
                r = (r + row[i]) & 0xffu; g = (g + row[i + 1]) & 0xffu; b = (b + row[i + 2]) & 0xffu;
                row[i] = (uint8_t)r; row[i + 1] = (uint8_t)g; row[i + 2] = (uint8_t)b;       // (the next row may be Up / Paeth)
                drow[i] = (uint8_t)b; drow[i + 1] = (uint8_t)g; drow[i + 2] = (uint8_t)r;
            }
            prev = row;
            continue;
        }
        switch (row[-1]) {
        case 0: break;
        case 1:
            for (size_t i = bpp; i < rowb; ++i) row[i] = (uint8_t)(row[i] + row[i - bpp]);
            break;
        case 2:
            for (size_t i = 0; i < rowb; ++i) row[i] = (uint8_t)(row[i] + prev[i]);
            break;
        case 3:
            for (size_t i = 0; i < (size_t)bpp; ++i) row[i] = (uint8_t)(row[i] + (prev[i] >> 1));
            for (size_t i = bpp; i < rowb; ++i) row[i] = (uint8_t)(row[i] + ((row[i - bpp] + prev[i]) >> 1));
            break;
        case 4:
            for (size_t i = 0; i < (size_t)bpp; ++i) row[i] = (uint8_t)(row[i] + prev[i]);
            for (size_t i = bpp; i < rowb; ++i) row[i] = (uint8_t)(row[i] + paeth(row[i - bpp], prev[i], prev[i - bpp]));
            break;
        default:
            err = "PNG: unknown filter type";
            return false;
        }
        if (drow)
            for (size_t i = 0; i < rowb; i += 3) { drow[i] = row[i + 2]; drow[i + 1] = row[i + 1]; drow[i + 2] = row[i]; }
        prev = row;
    }
    return true;
}

}  // namespace

// 0: decoded (out = h*w*3 bytes BGR, cv2.imread's IMREAD_COLOR), 1: corrupt / unreadable (err set), 2: a valid-looking
// PNG of a kind this reader does not take (16-bit, palette, interlaced)
int png_read_bgr(const uint8_t* file, size_t len, uint8_t* out, size_t cap, int* h_out, int* w_out, std::string& err)
{
    crc_init();
    static const uint8_t sig[8] = {0x89, 'P', 'N', 'G', '\r', '\n', 0x1a, '\n'};
    if (len < 8 + 25 + 12 || std::memcmp(file, sig, 8)) { err = "not a PNG file"; return 1; }
    size_t pos = 8;
    int w = 0, h = 0, ctype = -1;
    bool seen_ihdr = false, seen_iend = false;
    static thread_local std::vector<uint8_t> idat, raw;      // reused: a fresh 6 MB vector per frame is 1-2 ms of page faults
    idat.clear();
    const uint8_t* one_idat = nullptr;                       // a file with a single IDAT chunk is inflated in place
    size_t one_len = 0;
    int nidat = 0;
    while (pos + 12 <= len && !seen_iend) {
        const uint32_t n = rd_be32(file + pos);
        if (n > len - pos - 12) { err = "PNG: chunk runs past the end of the file"; return 1; }
        const uint8_t* const type = file + pos + 4;
        const uint8_t* const body = type + 4;
        if (crc32(0, type, n + 4) != rd_be32(body + n)) { err = "PNG: chunk CRC mismatch"; return 1; }
        if (!std::memcmp(type, "IHDR", 4)) {
            if (n != 13 || seen_ihdr) { err = "PNG: bad IHDR"; return 1; }
            seen_ihdr = true;
            w = (int)rd_be32(body); h = (int)rd_be32(body + 4);
            ctype = body[9];
            if (w <= 0 || h <= 0 || w > (1 << 24) || h > (1 << 24)) { err = "PNG: bad dimensions"; return 1; }
            if (body[10] || body[11]) { err = "PNG: unknown compression or filter method"; return 1; }
            if (body[8] != 8 || body[12] != 0 || !(ctype == 0 || ctype == 2 || ctype == 4 || ctype == 6)) return 2;
        } else if (!std::memcmp(type, "IDAT", 4)) {
            if (!seen_ihdr) { err = "PNG: IDAT before IHDR"; return 1; }
            if (++nidat == 1) { one_idat = body; one_len = n; }                       // the only chunk so far: no copy yet
            else {
                if (nidat == 2) idat.assign(one_idat, one_idat + one_len);
                idat.insert(idat.end(), body, body + n);
            }
        } else if (!std::memcmp(type, "IEND", 4)) {
            seen_iend = true;
        } else if (!(type[0] & 0x20)) {
            if (!std::memcmp(type, "PLTE", 4)) { /* allowed with colour types 2 and 6 as a suggestion */ }
            else { err = "PNG: unknown critical chunk"; return 1; }
        }
        pos += 12 + (size_t)n;
    }
    const uint8_t* const zdata = nidat >= 2 ? idat.data() : one_idat;
    const size_t zlen = nidat >= 2 ? idat.size() : one_len;
    if (!seen_ihdr || !seen_iend || zlen == 0) { err = "PNG: missing IHDR, IDAT or IEND"; return 1; }
    if (h_out) *h_out = h;
    if (w_out) *w_out = w;
    const size_t need = (size_t)h * w * 3;
    if (!out) return 0;                                   // size query
    if (cap < need) { err = "PNG: output buffer too small"; return 1; }
    const int bpp = ctype == 0 ? 1 : ctype == 4 ? 2 : ctype == 2 ? 3 : 4;
    // deflate cannot expand by more than 1032:1: a header that promises more than the IDAT data could hold is damage (or
    // an attempt to make the reader allocate terabytes)
    if ((size_t)h * ((size_t)w * bpp + 1) > zlen * 1032 + 1024) { err = "PNG: image larger than its data can be"; return 1; }
    raw.resize((size_t)h * ((size_t)w * bpp + 1) + 8);                    // + slack for the word-wise match copy
    if (!zlib_decompress(zdata, zlen, raw.data(), raw.size() - 8, err)) return 1;
    if (!unfilter(raw.data(), h, w, bpp, out, err)) return 1;
    if (bpp != 3)
    for (int y = 0; y < h; ++y) {
        const uint8_t* s = raw.data() + (size_t)y * ((size_t)w * bpp + 1) + 1;
        uint8_t* d = out + (size_t)y * w * 3;
        if (bpp >= 3)
            for (int x = 0; x < w; ++x, s += bpp, d += 3) { d[0] = s[2]; d[1] = s[1]; d[2] = s[0]; }     // RGB(A) -> BGR
        else
            for (int x = 0; x < w; ++x, s += bpp, d += 3) d[0] = d[1] = d[2] = s[0];                    // grey(+alpha)
    }
    return 0;
}

// test hook: the inflate alone (zlib-wrapped stream -> exactly out_len bytes)
int zlib_decompress_exact(const uint8_t* in, size_t n, uint8_t* out, size_t out_len, std::string& err)
{
    std::vector<uint8_t> tmp(out_len + 8);
    if (!zlib_decompress(in, n, tmp.data(), out_len, err)) return 1;
    std::memcpy(out, tmp.data(), out_len);
    return 0;
}

}  // namespace uva
