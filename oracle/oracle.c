/*
 * oracle.c -- CPU restatement of the per-frame super-resolution hot path of
 * davlee1972/upscale_video.  TEST INFRASTRUCTURE ONLY.
 *
 * Only tests/, __graft_entry__.smoke() and bench.py's cpu_baseline leg may load
 * this library, and only as the checker.  The product (upscale_video_amd/) never
 * links, imports or executes anything under oracle/.
 *
 * PARITY UNPINNED at the ncnn boundary (layer arithmetic); host logic pinned, see below.  The arithmetic of this path lives in the third-party,
 * un-vendored, un-pinned `ncnn_vulkan` wheel (reference README.md:29) and in
 * opencv-python; neither is installed here nor installable (no network), and the
 * reference holds no tests, golden vectors or fixtures for the path (SURVEY.md
 * section 4).  This file therefore restates ncnn's *published* layer semantics
 * (github.com/Tencent/ncnn, src/layer/{convolution,prelu,pixelshuffle,interp,
 * binaryop}.cpp, src/mat_pixel.cpp, src/modelbin.cpp, src/paramdict.cpp) and
 * OpenCV's convertTo(CV_8U) rounding, anchored on the reference's own call sites:
 *
 *   upscale/upscale_processing.py:65-73    net construction, .param/.bin load
 *   upscale/upscale_processing.py:263-273  imread(BGR) -> from_pixels -> x*(1/255)
 *   upscale/upscale_processing.py:278-288  extract -> transpose*255 -> imwrite
 *   upscale/upscale_processing.py:395-477  process_tile (960 px tile, 10 px border)
 *   upscale/upscale_processing.py:480-542  upscale_image (float64 canvas, imwrite)
 *   models/2x_Compact_Pretrain.param:3-42, models/4x_Compact_Pretrain.param:3-42,
 *   models/1x_HurrDeblur_SubCompact_nf24-nc8_244k_net_g.param:3-26  (graphs)
 *
 * It is cross-checked in this container against an independent torch-CPU
 * evaluation of the same .param/.bin (oracle/independent_check.py) and by loader
 * known-answer tests (byte-exact consumption of every .bin).
 *
 * Since round 6 the HOST LOGIC restated here (uvo_upscale_image_u8's 960/10
 * windows and paste, uvo_apply_model_u8's pre/post steps, the u8 conversion) IS
 * pinned by the reference's own code, executed: oracle/ref_host_fixtures.py
 * imports /root/reference/upscale/upscale_processing.py unchanged under stand-in
 * cv2 / ncnn_vulkan / wakepy modules, runs its upscale_image / process_tile /
 * apply_model and writes tests/golden/ref_host.npz; tests/test_ref_host.py holds
 * this file to those frames (equal except .5 ties).  What stays unpinned is the
 * layer arithmetic (ncnn's published semantics, the wheel being absent).
 *
 * Arithmetic: fp32 throughout, one rounding per multiply and per add (build with
 * -ffp-contract=off), accumulation order bias, then ci-major / ky / kx.
 */
#define _GNU_SOURCE
#include <math.h>
#include <stdint.h>
#include <stdio.h>
#include <stdlib.h>
#include <string.h>

#ifdef _OPENMP
#include <omp.h>
#endif

#define UVO_MAX_LAYERS 2048
#define UVO_MAX_BLOBS 4
#define UVO_NAME 64

enum {
    UVO_F16_STORAGE = 1, /* emulate the HIP path's storage precision: fp16 weights,
                            fp16 activations after every PReLU (fp32 accumulate) */
    UVO_F16_INPUT = 4,   /* with UVO_F16_STORAGE: the input blob itself is rounded to fp16 -- what the HIP path's float
                            (Extractor) route does with the caller's normalised ncnn::Mat (head_kernel<NF, 1>); the u8
                            route feeds the integer pixel values, which are exact */
    UVO_WINOGRAD_F23 = 2, /* with UVO_F16_STORAGE: the 64 -> 64 trunk convolutions that the HIP path runs
                            as fused pairs are evaluated as 1-D Winograd F(2,3) along x with the kernel's
                            rounding points (trunkw_kernel, csrc/uva_wino.hip.h): transformed inputs and
                            transformed weights rounded to fp16, products and sums in fp32 */
    UVO_PRELU_F16 = 8,   /* with UVO_WINOGRAD_F23: the PReLU behind such a convolution on fp16 values, as trunkw_kernel's
                            TW_ACT_F16 modes do it (csrc/uva_wino.h): the sum rounded to fp16, times the slope rounded to
                            fp16, product rounded to fp16, then max (slope <= 1) or min (slope > 1) of the two */
};

typedef enum {
    L_INPUT, L_SPLIT, L_CONV, L_PRELU, L_PIXELSHUFFLE, L_INTERP, L_BINARYOP, L_UNSUPPORTED
} ltype;

typedef struct {
    ltype type;
    char type_name[UVO_NAME];
    char name[UVO_NAME];
    int nin, nout;
    char in[UVO_MAX_BLOBS][UVO_NAME];
    char out[UVO_MAX_BLOBS][UVO_NAME];
    /* Convolution (ncnn convolution.cpp load_param ids) */
    int num_output;       /* 0= */
    int kernel;           /* 1= */
    int pad;              /* 4= */
    int bias_term;        /* 5= */
    int weight_data_size; /* 6= */
    int cin;
    uint32_t tag;         /* .bin flag word preceding the weights */
    float* w;             /* OIHW fp32 (expanded from fp16 when tagged) */
    float* w16;           /* same, each value rounded through fp16 */
    float* bias;
    /* PReLU */
    int num_slope;
    float* slope;
    /* PixelShuffle */
    int upscale;
    /* Interp */
    int resize_type;
    float hscale, wscale;
    /* BinaryOp */
    int op_type;
} layer;

typedef struct uvo_model {
    int nlayers, nblobs_decl;
    layer* L;
    size_t bin_size, bin_consumed;
    int scale; /* product of pixel-shuffle factors */
    int nf;    /* trunk width */
    int nconv;
} uvo_model;

/* ---------------- fp16 <-> fp32 (IEEE binary16, round-to-nearest-even) -------- */

static float half_bits_to_float(uint16_t h)
{
    uint32_t sign = (uint32_t)(h & 0x8000u) << 16;
    uint32_t exp = (h >> 10) & 0x1f;
    uint32_t man = h & 0x3ffu;
    uint32_t f;
    if (exp == 0) {
        if (man == 0) {
            f = sign;
        } else { /* subnormal */
            int e = -1;
            do { man <<= 1; ++e; } while (!(man & 0x400u));
            man &= 0x3ffu;
            f = sign | ((uint32_t)(127 - 15 - e) << 23) | (man << 13);
        }
    } else if (exp == 31) {
        f = sign | 0x7f800000u | (man << 13);
    } else {
        f = sign | ((exp + 127 - 15) << 23) | (man << 13);
    }
    float out;
    memcpy(&out, &f, 4);
    return out;
}

static uint16_t float_to_half_bits(float x)
{
    uint32_t f;
    memcpy(&f, &x, 4);
    uint32_t sign = (f >> 16) & 0x8000u;
    uint32_t a = f & 0x7fffffffu;
    if (a >= 0x7f800000u) /* inf / nan */
        return (uint16_t)(sign | 0x7c00u | ((a > 0x7f800000u) ? 0x200u : 0));
    if (a >= 0x477ff000u) /* rounds to >= 65520 -> inf */
        return (uint16_t)(sign | 0x7c00u);
    if (a < 0x33000001u) /* < 2^-25 (or == 2^-25 ties to even 0) */
        return (uint16_t)sign;
    int e = (int)(a >> 23) - 127;
    uint32_t m = (a & 0x7fffffu) | 0x800000u;
    int shift;
    uint32_t hexp;
    if (e < -14) { shift = 13 + (-14 - e); hexp = 0; }
    else { shift = 13; hexp = (uint32_t)(e + 15); }
    uint32_t q = m >> shift;
    uint32_t rem = m & ((1u << shift) - 1);
    uint32_t halfway = 1u << (shift - 1);
    if (rem > halfway || (rem == halfway && (q & 1))) ++q;
    uint32_t h;
    if (hexp == 0) h = q; /* subnormal (may carry into exp=1, which is correct) */
    else h = ((hexp - 1) << 10) + q; /* q carries the implicit bit */
    return (uint16_t)(sign | h);
}

float uvo_round_f16(float x) { return half_bits_to_float(float_to_half_bits(x)); }

/* ---------------- .param parser (ncnn text format, magic 7767517) ------------- */

static ltype classify(const char* t)
{
    if (!strcmp(t, "Input")) return L_INPUT;
    if (!strcmp(t, "Split")) return L_SPLIT;
    if (!strcmp(t, "Convolution")) return L_CONV;
    if (!strcmp(t, "PReLU")) return L_PRELU;
    if (!strcmp(t, "PixelShuffle")) return L_PIXELSHUFFLE;
    if (!strcmp(t, "Interp")) return L_INTERP;
    if (!strcmp(t, "BinaryOp")) return L_BINARYOP;
    return L_UNSUPPORTED;
}

static void seterr(char* err, int errlen, const char* fmt, const char* a)
{
    if (err && errlen > 0) snprintf(err, (size_t)errlen, fmt, a);
}

void uvo_free(uvo_model* m)
{
    if (!m) return;
    for (int i = 0; i < m->nlayers; ++i) {
        free(m->L[i].w); free(m->L[i].w16); free(m->L[i].bias); free(m->L[i].slope);
    }
    free(m->L);
    free(m);
}

static int parse_param(uvo_model* m, const char* path, char* err, int errlen)
{
    FILE* f = fopen(path, "r");
    if (!f) { seterr(err, errlen, "cannot open %s", path); return -1; }
    int magic = 0;
    if (fscanf(f, "%d", &magic) != 1 || magic != 7767517) {
        seterr(err, errlen, "bad magic in %s", path); fclose(f); return -1;
    }
    int nl = 0, nb = 0;
    if (fscanf(f, "%d %d", &nl, &nb) != 2 || nl <= 0 || nl > UVO_MAX_LAYERS) {
        seterr(err, errlen, "bad layer count in %s", path); fclose(f); return -1;
    }
    m->nlayers = nl; m->nblobs_decl = nb;
    m->L = (layer*)calloc((size_t)nl, sizeof(layer));
    char line[8192];
    if (!fgets(line, sizeof line, f)) { /* rest of count line */ }
    for (int i = 0; i < nl; ++i) {
        if (!fgets(line, sizeof line, f)) {
            seterr(err, errlen, "truncated %s", path); fclose(f); return -1;
        }
        layer* L = &m->L[i];
        L->kernel = 1; L->upscale = 1; L->hscale = 1.f; L->wscale = 1.f;
        L->resize_type = 0; L->op_type = 0;
        char* save = NULL;
        char* tok = strtok_r(line, " \t\r\n", &save);
        if (!tok) { --i; continue; }
        snprintf(L->type_name, UVO_NAME, "%s", tok);
        L->type = classify(tok);
        tok = strtok_r(NULL, " \t\r\n", &save);
        snprintf(L->name, UVO_NAME, "%s", tok ? tok : "");
        tok = strtok_r(NULL, " \t\r\n", &save); L->nin = tok ? atoi(tok) : 0;
        tok = strtok_r(NULL, " \t\r\n", &save); L->nout = tok ? atoi(tok) : 0;
        for (int k = 0; k < L->nin; ++k) {
            tok = strtok_r(NULL, " \t\r\n", &save);
            if (k < UVO_MAX_BLOBS) snprintf(L->in[k], UVO_NAME, "%s", tok ? tok : "");
        }
        for (int k = 0; k < L->nout; ++k) {
            tok = strtok_r(NULL, " \t\r\n", &save);
            if (k < UVO_MAX_BLOBS) snprintf(L->out[k], UVO_NAME, "%s", tok ? tok : "");
        }
        while ((tok = strtok_r(NULL, " \t\r\n", &save))) {
            char* eq = strchr(tok, '=');
            if (!eq) continue;
            int id = atoi(tok);
            const char* v = eq + 1;
            if (id <= -23300) continue; /* array params: not used by the supported layers */
            switch (L->type) {
            case L_CONV:
                if (id == 0) L->num_output = atoi(v);
                else if (id == 1) L->kernel = atoi(v);
                else if (id == 4) L->pad = atoi(v);
                else if (id == 5) L->bias_term = atoi(v);
                else if (id == 6) L->weight_data_size = atoi(v);
                break;
            case L_PRELU: if (id == 0) L->num_slope = atoi(v); break;
            case L_PIXELSHUFFLE: if (id == 0) L->upscale = atoi(v); break;
            case L_INTERP:
                if (id == 0) L->resize_type = atoi(v);
                else if (id == 1) L->hscale = (float)atof(v);
                else if (id == 2) L->wscale = (float)atof(v);
                break;
            case L_BINARYOP: if (id == 0) L->op_type = atoi(v); break;
            default: break;
            }
        }
    }
    fclose(f);
    return 0;
}

/* .bin stream (ncnn modelbin.cpp ModelBinFromDataReader::load):
 *   type 0 (Convolution weights): u32 flag; 0x01306B47 -> n fp16 values, padded to a
 *   4-byte boundary; 0 -> n raw fp32.  type 1 (bias, PReLU slopes): n raw fp32, no flag. */
static int load_bin(uvo_model* m, const char* path, char* err, int errlen)
{
    FILE* f = fopen(path, "rb");
    if (!f) { seterr(err, errlen, "cannot open %s", path); return -1; }
    fseek(f, 0, SEEK_END);
    long sz = ftell(f);
    fseek(f, 0, SEEK_SET);
    uint8_t* buf = (uint8_t*)malloc((size_t)sz);
    if (fread(buf, 1, (size_t)sz, f) != (size_t)sz) {
        seterr(err, errlen, "short read %s", path); fclose(f); free(buf); return -1;
    }
    fclose(f);
    m->bin_size = (size_t)sz;
    size_t off = 0;
    for (int i = 0; i < m->nlayers; ++i) {
        layer* L = &m->L[i];
        if (L->type == L_CONV) {
            int n = L->weight_data_size;
            int k2 = L->kernel * L->kernel;
            if (n <= 0 || L->num_output <= 0 || n % (L->num_output * k2)) {
                seterr(err, errlen, "bad conv sizes in %s", L->name); free(buf); return -1;
            }
            L->cin = n / (L->num_output * k2);
            if (off + 4 > (size_t)sz) goto trunc;
            memcpy(&L->tag, buf + off, 4); off += 4;
            L->w = (float*)malloc(sizeof(float) * (size_t)n);
            L->w16 = (float*)malloc(sizeof(float) * (size_t)n);
            if (L->tag == 0x01306B47u) {
                size_t bytes = ((size_t)n * 2 + 3) & ~(size_t)3;
                if (off + bytes > (size_t)sz) goto trunc;
                for (int k = 0; k < n; ++k) {
                    uint16_t h; memcpy(&h, buf + off + 2 * (size_t)k, 2);
                    L->w[k] = half_bits_to_float(h);
                }
                off += bytes;
            } else if (L->tag == 0) {
                if (off + (size_t)n * 4 > (size_t)sz) goto trunc;
                memcpy(L->w, buf + off, (size_t)n * 4); off += (size_t)n * 4;
            } else {
                seterr(err, errlen, "unsupported weight tag in %s", L->name); free(buf); return -1;
            }
            for (int k = 0; k < n; ++k) L->w16[k] = uvo_round_f16(L->w[k]);
            L->bias = (float*)calloc((size_t)L->num_output, sizeof(float));
            if (L->bias_term) {
                if (off + (size_t)L->num_output * 4 > (size_t)sz) goto trunc;
                memcpy(L->bias, buf + off, (size_t)L->num_output * 4);
                off += (size_t)L->num_output * 4;
            }
        } else if (L->type == L_PRELU) {
            if (off + (size_t)L->num_slope * 4 > (size_t)sz) goto trunc;
            L->slope = (float*)malloc(sizeof(float) * (size_t)L->num_slope);
            memcpy(L->slope, buf + off, (size_t)L->num_slope * 4);
            off += (size_t)L->num_slope * 4;
        }
    }
    m->bin_consumed = off;
    free(buf);
    return 0;
trunc:
    seterr(err, errlen, "truncated weights %s", path);
    free(buf);
    return -1;
}

uvo_model* uvo_load(const char* param_path, const char* bin_path, char* err, int errlen)
{
    uvo_model* m = (uvo_model*)calloc(1, sizeof *m);
    if (parse_param(m, param_path, err, errlen)) { uvo_free(m); return NULL; }
    for (int i = 0; i < m->nlayers; ++i)
        if (m->L[i].type == L_UNSUPPORTED) {
            seterr(err, errlen, "unsupported layer type %s", m->L[i].type_name);
            uvo_free(m); return NULL;
        }
    if (load_bin(m, bin_path, err, errlen)) { uvo_free(m); return NULL; }
    m->scale = 1;
    for (int i = 0; i < m->nlayers; ++i) {
        if (m->L[i].type == L_PIXELSHUFFLE) m->scale *= m->L[i].upscale;
        if (m->L[i].type == L_CONV) { if (!m->nconv) m->nf = m->L[i].num_output; m->nconv++; }
    }
    return m;
}

int uvo_scale(const uvo_model* m) { return m->scale; }
int uvo_nf(const uvo_model* m) { return m->nf; }
int uvo_num_conv(const uvo_model* m) { return m->nconv; }
size_t uvo_bin_size(const uvo_model* m) { return m->bin_size; }
size_t uvo_bin_consumed(const uvo_model* m) { return m->bin_consumed; }
int uvo_num_layers(const uvo_model* m) { return m->nlayers; }

static const layer* nth_conv(const uvo_model* m, int idx)
{
    for (int i = 0, k = 0; i < m->nlayers; ++i)
        if (m->L[i].type == L_CONV && k++ == idx) return &m->L[i];
    return NULL;
}

/* KAT accessors: conv idx -> cin, cout, tag, weights (OIHW), bias */
int uvo_conv_info(const uvo_model* m, int idx, int* cin, int* cout, uint32_t* tag)
{
    const layer* L = nth_conv(m, idx);
    if (!L) return -1;
    *cin = L->cin; *cout = L->num_output; *tag = L->tag;
    return 0;
}
const float* uvo_conv_weights(const uvo_model* m, int idx) { const layer* L = nth_conv(m, idx); return L ? L->w : NULL; }
const float* uvo_conv_bias(const uvo_model* m, int idx) { const layer* L = nth_conv(m, idx); return L ? L->bias : NULL; }
const float* uvo_prelu_slopes(const uvo_model* m, int idx, int* n)
{
    for (int i = 0, k = 0; i < m->nlayers; ++i)
        if (m->L[i].type == L_PRELU && k++ == idx) { *n = m->L[i].num_slope; return m->L[i].slope; }
    return NULL;
}

/* ---------------- layers (planar CHW fp32, like ncnn::Mat) -------------------- */

typedef struct { int c, h, w; float* d; char name[UVO_NAME]; int refs; } blob;

/* one output row of up to 4 output channels; the vector clones only change how many x
 * are processed per instruction, never the per-element operation order */
#if defined(__x86_64__) && defined(__GNUC__) && !defined(__clang__)
__attribute__((target_clones("default", "avx2", "avx512f")))
#endif
static void conv_rowblock(const layer* L, const float* wt, const float* pin, float* outd,
                          int o0, int y, int C, int O, int K, int PH, int PW, int OH, int OW)
{
    const int on = (O - o0) < 4 ? (O - o0) : 4;
    float* acc[4];
    for (int j = 0; j < 4; ++j)
        acc[j] = outd + ((size_t)(o0 + (j < on ? j : 0)) * OH + y) * OW;
    for (int j = 0; j < on; ++j) {
        const float b = L->bias[o0 + j];
        for (int x = 0; x < OW; ++x) acc[j][x] = b;
    }
    for (int ci = 0; ci < C; ++ci) {
        for (int ky = 0; ky < K; ++ky) {
            const float* irow = pin + ((size_t)ci * PH + y + ky) * PW;
            for (int kx = 0; kx < K; ++kx) {
                const float* ip = irow + kx;
                if (on == 4) {
                    const float w0 = wt[(((size_t)(o0 + 0) * C + ci) * K + ky) * K + kx];
                    const float w1 = wt[(((size_t)(o0 + 1) * C + ci) * K + ky) * K + kx];
                    const float w2 = wt[(((size_t)(o0 + 2) * C + ci) * K + ky) * K + kx];
                    const float w3 = wt[(((size_t)(o0 + 3) * C + ci) * K + ky) * K + kx];
                    float *restrict a0 = acc[0], *restrict a1 = acc[1];
                    float *restrict a2 = acc[2], *restrict a3 = acc[3];
                    for (int x = 0; x < OW; ++x) {
                        const float v = ip[x];
                        a0[x] += v * w0; a1[x] += v * w1; a2[x] += v * w2; a3[x] += v * w3;
                    }
                } else {
                    for (int j = 0; j < on; ++j) {
                        const float wj = wt[(((size_t)(o0 + j) * C + ci) * K + ky) * K + kx];
                        float* restrict a = acc[j];
                        for (int x = 0; x < OW; ++x) a[x] += ip[x] * wj;
                    }
                }
            }
        }
    }
}

/* ncnn convolution.cpp: zero pad `pad` on every side, stride 1, dilation 1,
 * out[co][y][x] = bias[co] + sum_ci sum_ky sum_kx in[ci][y+ky-pad][x+kx-pad]*w[co][ci][ky][kx] */
static void conv2d(const layer* L, const float* wt, const blob* in, blob* out, int nthreads)
{
    const int C = L->cin, O = L->num_output, K = L->kernel, P = L->pad;
    const int H = in->h, W = in->w;
    const int OH = H + 2 * P - K + 1, OW = W + 2 * P - K + 1;
    out->c = O; out->h = OH; out->w = OW;
    out->d = (float*)malloc(sizeof(float) * (size_t)O * OH * OW);
    /* zero-padded copy of the input so the inner loop has no branches */
    const int PH = H + 2 * P, PW = W + 2 * P;
    float* pin = (float*)calloc((size_t)C * PH * PW, sizeof(float));
    for (int c = 0; c < C; ++c)
        for (int y = 0; y < H; ++y)
            memcpy(pin + ((size_t)c * PH + y + P) * PW + P, in->d + ((size_t)c * H + y) * W,
                   sizeof(float) * (size_t)W);
    const int OB = 4; /* output channels per register block */
    const int nblk = (O + OB - 1) / OB;
    (void)nthreads;
#ifdef _OPENMP
#pragma omp parallel for collapse(2) schedule(static) num_threads(nthreads > 0 ? nthreads : 1)
#endif
    for (int ob = 0; ob < nblk; ++ob)
        for (int y = 0; y < OH; ++y)
            conv_rowblock(L, wt, pin, out->d, ob * OB, y, C, O, K, PH, PW, OH, OW);
    free(pin);
}

/* The same convolution (3x3, pad 1, stride 1) as 1-D Winograd F(2,3) along x, the way trunkw_kernel evaluates
 * it.  Output columns come in pairs (xa, xa+1), xa = align + 2q (align = -1 for the first layer of a fused
 * pair, 0 for the second: csrc/uva_wino.hip.h); with d(x) the fp16 input activation (0 outside the plane):
 *   V0 = f16(d(xa-1) - d(xa+1))  V1 = f16(d(xa) + d(xa+1))  V2 = f16(d(xa+1) - d(xa))  V3 = f16(d(xa) - d(xa+2))
 *   U0 = f16(g0)  U1 = f16((g0+g1+g2)/2)  U2 = f16((g0-g1+g2)/2)  U3 = f16(g2)     (g = the row's three taps)
 *   Mj = sum over ci, ky of Uj * Vj (fp32), M1's sum starting at the bias;  out(xa) = (M0 + M1) + M2,  out(xa+1) = (M1 - M2) - M3.
 * In exact arithmetic this IS conv2d (Lavin & Gray 2015, F(2,3)); what differs is where fp16 rounds. */
static void conv2d_wino_f23(const layer* L, const blob* in, blob* out, int align, int nthreads)
{
    const int C = L->cin, O = L->num_output, H = in->h, W = in->w;
    const int NP = (W - 1 - align) / 2 + 1;           /* pairs covering columns align .. W-1 */
    const int PH = H + 2;
    out->c = O; out->h = H; out->w = W;
    out->d = (float*)malloc(sizeof(float) * (size_t)O * H * W);
    float* V = (float*)calloc((size_t)4 * C * PH * NP, sizeof(float));      /* [j][ci][y+1][q], rows -1 and H zero */
    float* U = (float*)malloc(sizeof(float) * (size_t)4 * O * C * 3);       /* [j][co][ci][ky] */
    for (int co = 0; co < O; ++co)
        for (int ci = 0; ci < C; ++ci)
            for (int ky = 0; ky < 3; ++ky) {
                const float* g = L->w + (((size_t)co * C + ci) * 3 + ky) * 3;
                const double g0 = g[0], g1 = g[1], g2 = g[2];
                const size_t k = ((size_t)co * C + ci) * 3 + ky, js = (size_t)O * C * 3;
                U[0 * js + k] = uvo_round_f16(g[0]);
                U[1 * js + k] = uvo_round_f16((float)(0.5 * (g0 + g1 + g2)));
                U[2 * js + k] = uvo_round_f16((float)(0.5 * (g0 - g1 + g2)));
                U[3 * js + k] = uvo_round_f16(g[2]);
            }
    (void)nthreads;
#ifdef _OPENMP
#pragma omp parallel for collapse(2) schedule(static) num_threads(nthreads > 0 ? nthreads : 1)
#endif
    for (int ci = 0; ci < C; ++ci)
        for (int y = 0; y < H; ++y) {
            const float* row = in->d + ((size_t)ci * H + y) * W;
#define D_(x) (((x) >= 0 && (x) < W) ? row[(x)] : 0.f)
            for (int q = 0; q < NP; ++q) {
                const int xa = align + 2 * q;
                const float d0 = D_(xa - 1), d1 = D_(xa), d2 = D_(xa + 1), d3 = D_(xa + 2);
                const size_t k = ((size_t)ci * PH + y + 1) * NP + q, js = (size_t)C * PH * NP;
                V[0 * js + k] = uvo_round_f16(d0 - d2);
                V[1 * js + k] = uvo_round_f16(d1 + d2);
                V[2 * js + k] = uvo_round_f16(d2 - d1);
                V[3 * js + k] = uvo_round_f16(d1 - d3);
            }
#undef D_
        }
#ifdef _OPENMP
#pragma omp parallel for collapse(2) schedule(static) num_threads(nthreads > 0 ? nthreads : 1)
#endif
    for (int co = 0; co < O; ++co)
        for (int y = 0; y < H; ++y) {
            float* M = (float*)calloc((size_t)4 * NP, sizeof(float));
            const float b = L->bias[co];
            for (int q = 0; q < NP; ++q) M[NP + q] = b;     /* M1 enters both results with +: its sum starts at the bias */
            for (int j = 0; j < 4; ++j) {
                float* restrict mj = M + (size_t)j * NP;
                for (int ci = 0; ci < C; ++ci)
                    for (int ky = 0; ky < 3; ++ky) {
                        const float u = U[(size_t)j * O * C * 3 + ((size_t)co * C + ci) * 3 + ky];
                        const float* restrict v = V + (size_t)j * C * PH * NP + ((size_t)ci * PH + y + ky) * NP;
                        for (int q = 0; q < NP; ++q) mj[q] += u * v[q];
                    }
            }
            float* o = out->d + ((size_t)co * H + y) * W;
            for (int q = 0; q < NP; ++q) {
                const int xa = align + 2 * q;
                const float m0 = M[q], m1 = M[NP + q], m2 = M[2 * NP + q], m3 = M[3 * (size_t)NP + q];
                if (xa >= 0) o[xa] = (m0 + m1) + m2;
                if (xa + 1 < W) o[xa + 1] = (m1 - m2) - m3;
            }
            free(M);
        }
    free(V); free(U);
}

/* ncnn prelu.cpp: x < 0 ? x * slope[c] : x (num_slope == channels here) */
static void prelu(const layer* L, blob* b, int f16)
{
    const size_t hw = (size_t)b->h * b->w;
    for (int c = 0; c < b->c; ++c) {
        const float s = L->num_slope > 1 ? L->slope[c] : L->slope[0];
        float* p = b->d + (size_t)c * hw;
        for (size_t i = 0; i < hw; ++i) {
            float v = p[i];
            if (v < 0.f) v *= s;
            p[i] = f16 ? uvo_round_f16(v) : v;
        }
    }
}

/* the same PReLU the way trunkw_kernel evaluates it on packed halves (UVO_PRELU_F16): x16 = f16(x), m = f16(x16 * f16(s)),
 * max(x16, m) for s <= 1 and min(x16, m) for s > 1 -- PReLU in both cases, with two roundings instead of one for x < 0.
 * (The kernel computes a channel with s > 1 negated and takes max as well: -max(-x16, -m) = min(x16, m), every step exact.) */
static void prelu_f16(const layer* L, blob* b)
{
    const size_t hw = (size_t)b->h * b->w;
    for (int c = 0; c < b->c; ++c) {
        const float s = L->num_slope > 1 ? L->slope[c] : L->slope[0];
        const float s16 = uvo_round_f16(s);
        float* p = b->d + (size_t)c * hw;
        for (size_t i = 0; i < hw; ++i) {
            const float x = uvo_round_f16(p[i]);
            const float m = uvo_round_f16(x * s16);     /* (the product of two halves is exact in fp32: one rounding) */
            p[i] = s <= 1.f ? (x > m ? x : m) : (x < m ? x : m);
        }
    }
}

/* ncnn pixelshuffle.cpp mode 0: out[c][h*r+i][w*r+j] = in[c*r*r + i*r + j][h][w] */
static void pixelshuffle(const layer* L, const blob* in, blob* out)
{
    const int r = L->upscale;
    out->c = in->c / (r * r); out->h = in->h * r; out->w = in->w * r;
    out->d = (float*)malloc(sizeof(float) * (size_t)out->c * out->h * out->w);
    for (int c = 0; c < out->c; ++c)
        for (int i = 0; i < r; ++i)
            for (int j = 0; j < r; ++j) {
                const float* s = in->d + (size_t)(c * r * r + i * r + j) * in->h * in->w;
                for (int h = 0; h < in->h; ++h)
                    for (int w = 0; w < in->w; ++w)
                        out->d[((size_t)c * out->h + h * r + i) * out->w + w * r + j] =
                            s[(size_t)h * in->w + w];
            }
}

/* ncnn interp.cpp resize_type 1 (nearest): out dims = (int)(dim * scale),
 * in_y = min((int)(y * (h / (float)outh)), h - 1) */
static int interp(const layer* L, const blob* in, blob* out)
{
    if (L->resize_type != 1) return -1;
    out->c = in->c; out->h = (int)(in->h * L->hscale); out->w = (int)(in->w * L->wscale);
    out->d = (float*)malloc(sizeof(float) * (size_t)out->c * out->h * out->w);
    const float hs = in->h / (float)out->h, ws = in->w / (float)out->w;
    for (int c = 0; c < in->c; ++c)
        for (int y = 0; y < out->h; ++y) {
            int iy = (int)(y * hs); if (iy > in->h - 1) iy = in->h - 1;
            for (int x = 0; x < out->w; ++x) {
                int ix = (int)(x * ws); if (ix > in->w - 1) ix = in->w - 1;
                out->d[((size_t)c * out->h + y) * out->w + x] =
                    in->d[((size_t)c * in->h + iy) * in->w + ix];
            }
        }
    return 0;
}

static blob* find_blob(blob* bl, int nb, const char* name)
{
    for (int i = 0; i < nb; ++i) if (!strcmp(bl[i].name, name)) return &bl[i];
    return NULL;
}

/* Runs the parsed graph.  If tap_conv >= 0, stops after convolution #tap_conv and its
 * PReLU (if one follows) and returns that activation instead of the net output. */
static int run_graph(const uvo_model* m, const float* in_chw, int h, int w, int flags,
                     int nthreads, int tap_conv, float** out_d, int* oc, int* oh, int* ow)
{
    const int f16 = (flags & UVO_F16_STORAGE) != 0;
    blob* bl = (blob*)calloc((size_t)m->nlayers * 2 + 4, sizeof(blob));
    int nb = 0, rc = -1, convs = 0, tapped = 0, last_wino = 0;
    blob* result = NULL;
    for (int i = 0; i < m->nlayers && !tapped; ++i) {
        const layer* L = &m->L[i];
        blob* a = L->nin > 0 ? find_blob(bl, nb, L->in[0]) : NULL;
        if (L->nin > 0 && !a) goto done;
        switch (L->type) {
        case L_INPUT: {
            blob* o = &bl[nb++];
            snprintf(o->name, UVO_NAME, "%s", L->out[0]);
            o->c = 3; o->h = h; o->w = w;
            o->d = (float*)malloc(sizeof(float) * 3 * (size_t)h * w);
            memcpy(o->d, in_chw, sizeof(float) * 3 * (size_t)h * w);
            if (f16 && (flags & UVO_F16_INPUT))
                for (size_t k = 0; k < 3 * (size_t)h * w; ++k) o->d[k] = uvo_round_f16(o->d[k]);
            break;
        }
        case L_SPLIT:
            for (int k = 0; k < L->nout; ++k) {
                blob* o = &bl[nb++];
                *o = *a;
                snprintf(o->name, UVO_NAME, "%s", L->out[k]);
                o->d = (float*)malloc(sizeof(float) * (size_t)a->c * a->h * a->w);
                memcpy(o->d, a->d, sizeof(float) * (size_t)a->c * a->h * a->w);
            }
            break;
        case L_CONV: {
            blob* o = &bl[nb++];
            snprintf(o->name, UVO_NAME, "%s", L->out[0]);
            if (a->c != L->cin) goto done;
            /* the HIP path fuses trunk convolutions (2k-1, 2k) into one launch; a debug tap on 2k-1 runs it alone,
             * through the direct kernel (uva_api.hip run_graph) */
            const int second = convs + (convs & 1);
            if (f16 && (flags & UVO_WINOGRAD_F23) && m->nf == 64 && L->kernel == 3 && L->cin == 64 && L->num_output == 64 &&
                convs >= 1 && second <= m->nconv - 2 && (tap_conv < 0 || second <= tap_conv)) {
                conv2d_wino_f23(L, a, o, (convs & 1) ? -1 : 0, nthreads);
                last_wino = 1;
            } else {
                conv2d(L, f16 ? L->w16 : L->w, a, o, nthreads);
                last_wino = 0;
            }
            free(a->d); a->d = NULL;
            if (convs == tap_conv && !(i + 1 < m->nlayers && m->L[i + 1].type == L_PRELU)) {
                result = o; tapped = 1;
            }
            ++convs;
            break;
        }
        case L_PRELU: { /* in-place in ncnn as well */
            if (last_wino && (flags & UVO_PRELU_F16)) prelu_f16(L, a);
            else prelu(L, a, f16);
            snprintf(a->name, UVO_NAME, "%s", L->out[0]);
            if (convs - 1 == tap_conv) { result = a; tapped = 1; }
            break;
        }
        case L_PIXELSHUFFLE: {
            blob* o = &bl[nb++];
            snprintf(o->name, UVO_NAME, "%s", L->out[0]);
            pixelshuffle(L, a, o);
            free(a->d); a->d = NULL;
            break;
        }
        case L_INTERP: {
            blob* o = &bl[nb++];
            snprintf(o->name, UVO_NAME, "%s", L->out[0]);
            if (interp(L, a, o)) goto done;
            free(a->d); a->d = NULL;
            break;
        }
        case L_BINARYOP: {
            blob* b2 = find_blob(bl, nb, L->in[1]);
            if (!b2 || L->op_type != 0 || a->c != b2->c || a->h != b2->h || a->w != b2->w) goto done;
            const size_t n = (size_t)a->c * a->h * a->w;
            for (size_t k = 0; k < n; ++k) a->d[k] = a->d[k] + b2->d[k]; /* ncnn binaryop.cpp ADD */
            snprintf(a->name, UVO_NAME, "%s", L->out[0]);
            break;
        }
        default: goto done;
        }
    }
    if (!result) result = find_blob(bl, nb, "output");
    if (!result || !result->d) goto done;
    *out_d = result->d; *oc = result->c; *oh = result->h; *ow = result->w;
    result->d = NULL;
    rc = 0;
done:
    for (int i = 0; i < nb; ++i) free(bl[i].d);
    free(bl);
    return rc;
}

/* ex.input("input", mat_in); ex.extract("output")  (upscale_processing.py:278-281, :450-453).
 * out_chw must hold 3 * (h*scale) * (w*scale) floats. */
int uvo_forward_f32(const uvo_model* m, const float* in_chw, int h, int w, float* out_chw,
                    int flags, int nthreads)
{
    float* d; int c, oh, ow;
    if (run_graph(m, in_chw, h, w, flags, nthreads, -1, &d, &c, &oh, &ow)) return -1;
    memcpy(out_chw, d, sizeof(float) * (size_t)c * oh * ow);
    free(d);
    return 0;
}

/* activation after convolution #conv_idx (+PReLU): out holds cout*h*w floats, CHW */
int uvo_forward_tap(const uvo_model* m, const float* in_chw, int h, int w, int conv_idx,
                    float* out, int flags, int nthreads)
{
    float* d; int c, oh, ow;
    if (run_graph(m, in_chw, h, w, flags, nthreads, conv_idx, &d, &c, &oh, &ow)) return -1;
    memcpy(out, d, sizeof(float) * (size_t)c * oh * ow);
    free(d);
    return 0;
}

/* ncnn mat_pixel.cpp from_pixels(PIXEL_BGR): byte k of each pixel -> plane k, no swap;
 * then substract_mean_normalize(mean=[], norm=[1/255.0]*3): x *= (float)(1/255.0)
 * (upscale_processing.py:265-273, :437-445). */
void uvo_from_pixels_normalize(const uint8_t* hwc, int h, int w, size_t row_stride, float* chw)
{
    const float norm = (float)(1 / 255.0);
    for (int c = 0; c < 3; ++c)
        for (int y = 0; y < h; ++y)
            for (int x = 0; x < w; ++x)
                chw[((size_t)c * h + y) * w + x] = (float)hwc[y * row_stride + (size_t)x * 3 + c] * norm;
}

/* out.transpose(1,2,0) * 255 in float32, then cv2.imwrite -> Mat::convertTo(CV_8U):
 * saturate_cast<uchar>(cvRound(v)), cvRound = round-half-to-even (lrint)
 * (upscale_processing.py:284-288, :462, :497, :519).  The float64 canvas of
 * upscale_image holds the float32 products exactly, so rounding the float is identical. */
static uint8_t quantize(float v)
{
    float s = v * 255.0f;
    float r = nearbyintf(s); /* default rounding mode: ties to even */
    if (!(r > 0.f)) return 0;
    if (r > 255.f) return 255;
    return (uint8_t)r;
}

void uvo_to_u8(const float* chw, int h, int w, uint8_t* hwc, size_t row_stride)
{
    for (int y = 0; y < h; ++y)
        for (int x = 0; x < w; ++x)
            for (int c = 0; c < 3; ++c)
                hwc[y * row_stride + (size_t)x * 3 + c] = quantize(chw[((size_t)c * h + y) * w + x]);
}

/* apply_model (upscale_processing.py:258-299): whole frame, no tiling */
int uvo_apply_model_u8(const uvo_model* m, const uint8_t* in, int h, int w, uint8_t* out,
                       int flags, int nthreads)
{
    const int s = m->scale;
    float* fin = (float*)malloc(sizeof(float) * 3 * (size_t)h * w);
    float* fout = (float*)malloc(sizeof(float) * 3 * (size_t)h * s * w * s);
    uvo_from_pixels_normalize(in, h, w, (size_t)w * 3, fin);
    int rc = uvo_forward_f32(m, fin, h, w, fout, flags, nthreads);
    if (!rc) uvo_to_u8(fout, h * s, w * s, out, (size_t)w * s * 3);
    free(fin); free(fout);
    return rc;
}

/* upscale_image + process_tile (upscale_processing.py:395-542): tile grid of
 * ceil(W/tile) x ceil(H/tile); each tile is extended by `border` px on every side that is
 * at least `border` px away from the image edge (:409-427), run through the net on its
 * own (zero padding at the extended tile's edges), scaled, and its core pasted (:464-477). */
int uvo_upscale_image_u8(const uvo_model* m, const uint8_t* in, int h, int w, int tile_size,
                         int border, uint8_t* out, int flags, int nthreads)
{
    const int s = m->scale;
    const int tiles_x = (w + tile_size - 1) / tile_size, tiles_y = (h + tile_size - 1) / tile_size;
    const size_t ostride = (size_t)w * s * 3;
    for (int ty = 0; ty < tiles_y; ++ty)
        for (int tx = 0; tx < tiles_x; ++tx) {
            const int y0 = ty * tile_size, x0 = tx * tile_size;
            const int y1 = y0 + tile_size < h ? y0 + tile_size : h;
            const int x1 = x0 + tile_size < w ? x0 + tile_size : w;
            const int by0 = y0 >= border ? border : 0, by1 = y1 <= h - border ? border : 0;
            const int bx0 = x0 >= border ? border : 0, bx1 = x1 <= w - border ? border : 0;
            const int th = (y1 + by1) - (y0 - by0), tw = (x1 + bx1) - (x0 - bx0);
            float* fin = (float*)malloc(sizeof(float) * 3 * (size_t)th * tw);
            float* fout = (float*)malloc(sizeof(float) * 3 * (size_t)th * s * tw * s);
            uvo_from_pixels_normalize(in + (size_t)(y0 - by0) * w * 3 + (size_t)(x0 - bx0) * 3,
                                      th, tw, (size_t)w * 3, fin);
            if (uvo_forward_f32(m, fin, th, tw, fout, flags, nthreads)) { free(fin); free(fout); return -1; }
            const int oh = th * s, ow = tw * s;
            for (int y = y0 * s; y < y1 * s; ++y)
                for (int x = x0 * s; x < x1 * s; ++x) {
                    const int sy = y - (y0 - by0) * s, sx = x - (x0 - bx0) * s;
                    for (int c = 0; c < 3; ++c)
                        out[y * ostride + (size_t)x * 3 + c] =
                            quantize(fout[((size_t)c * oh + sy) * ow + sx]);
                }
            free(fin); free(fout);
        }
    return 0;
}

int uvo_max_threads(void)
{
#ifdef _OPENMP
    return omp_get_max_threads();
#else
    return 1;
#endif
}
