/*
 * uva.h -- C ABI of libuva.so, the MI355X (gfx950) per-frame super-resolution engine that
 * replaces the `ncnn_vulkan` surface used by davlee1972/upscale_video's worker functions.
 *
 * Every entry point cites the reference call site (file:line under /root/reference) whose
 * ncnn call it replaces.  Plain pointers and sizes only; no torch / numpy types.
 * All functions returning int return 0 on success and non-zero on error (ncnn's
 * convention for load_param/load_model/extract); uva_last_error() gives the message of the
 * last failure on the calling thread.
 *
 * Threading: a uva_net belongs to one thread at a time (the reference holds one
 * process-global net per pool worker, upscale/upscale_processing.py:22,57,65).  HIP is
 * initialised lazily inside the calling process, so the library is safe to load in a
 * multiprocessing "spawn" worker (:321, :565).
 *
 * There is NO CPU path: with no usable HIP device every compute entry point fails.
 */
#ifndef UVA_H
#define UVA_H

#include <stddef.h>
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

#define UVA_ABI_VERSION 15  /* 2: + uva_net_submit_u8 / uva_net_collect_u8 / uva_host_alloc / uva_host_free;
                               3: + uva_get_gpu_pci_bus_id, uva_debug_trunk2_schedule;
                               4: + uva_denoise_u8, uva_debug_denoise_stage; generic graphs (4x_Valar_v1) load;
                               5: + uva_debug_sub10_rows;
                               6: + uva_net_submit_u8_png, uva_png_workspace_bytes, uva_png_assemble, uva_png_deflate_u8,
                                  uva_debug_png_deflate_host;
                               7: + uva_png_decode_bgr, uva_debug_zlib_decompress;
                               8: + uva_net_debug_generic_plan;
                               9: + uva_debug_generic_segments (generic graphs: the fused residual-dense-block kernels);
                               10: + uva_debug_generic_segments_planes (a frame's reference tiles through those kernels in one launch);
                               11: + uva_debug_generic_batches;
                               12: + uva_debug_trunkw_schedule (trunkw_kernel: fused trunk pairs as Winograd F(2,3));
                               13: + uva_denoise_u8_device, uva_denoise_synchronize;
                               14: + uva_net_device, uva_debug_sub5_rows (sub5_kernel: the 1x net as two launches of five layers);
                               15: + uva_net_process_u8_device_batch (several frames of one geometry per call; the 1x net takes up to
                                   eight per launch), uva_debug_sub10_rows_batch */

typedef struct uva_net uva_net;

/* ---- device enumeration ------------------------------------------------------------ */

/* ncnn.get_gpu_count()            test_gpus.py:47 */
int uva_get_gpu_count(void);
/* ncnn.get_default_gpu_index()    test_gpus.py:53 */
int uva_get_default_gpu_index(void);
/* ncnn.get_gpu_info(i).type() / .device_name()   test_gpus.py:59-66
 * *type: 0 discrete, 1 integrated, 2 virtual, 3 cpu (ncnn's enumeration). */
int uva_get_gpu_info(int index, int* type, char* name, size_t name_len);
/* PCI address "dddd:bb:dd.f" of HIP device `index` (ncnn's GpuInfo has no counterpart; the worker layer
 * uses it to find the GPU's NUMA node in sysfs and to place a worker's feeder threads and page-locked
 * buffers next to the GPU it was given by its position in the -g list, upscale/upscale_processing.py:59). */
int uva_get_gpu_pci_bus_id(int index, char* out, size_t out_len);
/* ncnn.destroy_gpu_instance()     upscale/upscale_processing.py:292, :458
 * Frees every cached device allocation of every live net (nets stay loadable). */
void uva_destroy_gpu_instance(void);

/* ---- net life cycle ---------------------------------------------------------------- */

/* ncnn.Net()                      upscale/upscale_processing.py:65 */
uva_net* uva_net_create(void);
/* net.opt.use_vulkan_compute = True; net.set_vulkan_device(gpus[gpu])   :67-68
 * device = HIP ordinal.  A negative index is rejected (no CPU path). */
int uva_net_set_device(uva_net* net, int device);
/* The HIP ordinal the net is bound to: what set_vulkan_device was given, 0 (ncnn's default GPU index,
 * test_gpus.py:53) if it never was.  The shim asks the library instead of remembering (uva_denoise_u8_device takes
 * the ordinal and refuses a net that sits elsewhere). */
int uva_net_device(const uva_net* net);
/* net.load_param(path)            :70   parses the ncnn text graph.  The SRVGGNetCompact
 * pattern (conv3x3+PReLU stack, conv3x3, PixelShuffle r, Interp nearest r, BinaryOp add) takes
 * the fused kernels; any other graph made of Input / Split / Convolution 3x3 or 1x1 (+ fused
 * LeakyReLU) / Concat / BinaryOp add / Eltwise sum / Interp nearest / PReLU / PixelShuffle --
 * 4x_Valar_v1, `-m r`, :913-916 -- takes the generic executor (ABI 4); a layer type outside that
 * list is an error naming it. */
int uva_net_load_param(uva_net* net, const char* param_path);
/* net.load_model(path)            :71   reads the .bin stream; every byte must be
 * consumed.  Weights are repacked for the MFMA kernels and uploaded on first use. */
int uva_net_load_model(uva_net* net, const char* bin_path);
/* ncnn::Net::~Net */
void uva_net_destroy(uva_net* net);

/* graph facts after load_param */
int uva_net_scale(const uva_net* net);        /* 1, 2 or 4 */
int uva_net_num_features(const uva_net* net); /* 64 or 24 */
int uva_net_num_convs(const uva_net* net);    /* 18 or 10 */

/* ---- inference --------------------------------------------------------------------- */

/* ex = net.create_extractor(); ex.input("input", mat_in); ret, mat_out = ex.extract("output");
 * np.array(mat_out)               upscale/upscale_processing.py:278-281, :450-453
 * in_chw : host f32 planar [3][h][w] (what Mat.from_pixels + substract_mean_normalize made)
 * out_chw: host f32 planar [3][h*s][w*s] (what np.array(mat_out) returns) */
int uva_net_extract_f32(uva_net* net, const float* in_chw, int h, int w, float* out_chw);

/* Fused frame call: from_pixels(PIXEL_BGR) + substract_mean_normalize + extract +
 * transpose*255 + cv2 convertTo(CV_8U), all on the device, i.e. the arithmetic of
 *   apply_model   upscale/upscale_processing.py:263-288   (tile_size <= 0: whole frame)
 *   upscale_image + process_tile   :395-519               (tile_size = 960, border = 10)
 * in : u8 HWC BGR [h][w][3], rows in_stride bytes apart
 * out: u8 HWC BGR [h*s][w*s][3], rows out_stride bytes apart
 * Host pointers; H2D/D2H staged through pinned buffers on the net's stream. */
int uva_net_process_u8(uva_net* net, const uint8_t* in, int h, int w, size_t in_stride,
                       uint8_t* out, size_t out_stride, int tile_size, int border);

/* Pipelined form of uva_net_process_u8 for callers that stream frames (SURVEY.md 8d "host-to-host
 * with stream overlap"): submit returns at once with a ticket >= 0 (-1 on error); the frame's H2D
 * copy, kernels and D2H copy run on three streams, so consecutive frames overlap.  At most 3 frames
 * may be in flight; tickets are collected in any order.  `in` must stay valid until submit returns
 * when it is pageable (it is staged) and until collect returns when it is pinned (uva_host_alloc,
 * hipHostMalloc, hipHostRegister: copied from directly); `out` must stay valid until collect
 * returns, which is when it holds the result.  uva_net_process_u8 == submit + collect. */
long long uva_net_submit_u8(uva_net* net, const uint8_t* in, int h, int w, size_t in_stride,
                            uint8_t* out, size_t out_stride, int tile_size, int border);
int uva_net_collect_u8(uva_net* net, long long ticket);
/* Page-locked host memory for frames handed to the two calls above (no staging copy). */
void* uva_host_alloc(size_t bytes);
void uva_host_free(void* p);

/* ---- the imwrite side on the device (csrc/uva_png.hip.h) -------------------------------- */

/* cv2.imwrite(frame.png) of the result frame (upscale_processing.py:288, :519), with the deflate work done on the GPU:
 * like uva_net_submit_u8, but the result frame stays in HBM, a kernel behind the net compresses it (Sub filter, fixed
 * Huffman tables, one deflate block per group of rows) and the copy engine brings the compressed blocks into `png_ws`:
 * page-locked memory (uva_host_alloc) of uva_png_workspace_bytes(h*scale, w*scale) bytes that must stay untouched until
 * uva_net_collect_u8(ticket) has returned (collect completes the workspace: a frame that compressed worse than the
 * one before it has its last bytes fetched there).  uva_png_assemble() -- host only, any thread -- then turns the workspace into
 * the bytes of the PNG file: *len receives the size; out may be NULL to ask for it (the call then fails after setting
 * *len).  h, w there are the RESULT frame's.  Any PNG reader decodes the file to exactly the frame
 * uva_net_submit_u8 would have returned.  uva_png_workspace_bytes() returns 0 for frames the encoder does not take
 * (wider than 16 383 pixels). */
long long uva_net_submit_u8_png(uva_net* net, const uint8_t* in, int h, int w, size_t in_stride, void* png_ws,
                                size_t png_ws_bytes, int tile_size, int border);
size_t uva_png_workspace_bytes(int h, int w);
int uva_png_assemble(const void* png_ws, int h, int w, uint8_t* out, size_t cap, size_t* len);
/* The same encoder for a frame in host memory (synchronous: H2D copy, kernel, wait): imwrite for frames that did not
 * come out of a net of this library, e.g. the denoise pass (upscale_processing.py:336-338). */
int uva_png_deflate_u8(int gpu, const uint8_t* bgr, int h, int w, size_t stride, void* png_ws, size_t png_ws_bytes);
/* cv2.imread(path) (upscale_processing.py:263, :487) for a file image already in memory, host only, any thread: a
 * from-scratch inflate + PNG un-filter straight into cv2's u8 HWC BGR layout, 2-3x a zlib-based reader.  8-bit RGB,
 * RGBA (alpha dropped), grey (replicated), non-interlaced.  out == NULL: only *h, *w are set.  Returns 0, 1 for a
 * corrupt file (chunk CRC, Adler-32 and stream length are all checked; uva_last_error says what), 2 for a PNG of
 * another kind (16-bit, palette, interlaced): use a general reader for those. */
int uva_png_decode_bgr(const uint8_t* file, size_t len, uint8_t* out, size_t cap, int* h, int* w);
/* Test hook: the inflate alone on a zlib-wrapped stream that must expand to exactly out_len bytes. */
int uva_debug_zlib_decompress(const uint8_t* in, size_t n, uint8_t* out, size_t out_len);
/* Test hook (host only): the kernel's arithmetic restated on the host, block for block and bit for bit, into an
 * ordinary buffer of uva_png_workspace_bytes(h, w) bytes. */
int uva_debug_png_deflate_host(const uint8_t* bgr, int h, int w, size_t stride, void* png_ws, size_t png_ws_bytes);

/* Same, with in/out already resident in this device's HBM (dense rows: in_stride = 3*w,
 * out_stride = 3*w*s unless stated).  Asynchronous on the net's stream; follow with
 * uva_net_synchronize().  This is the call bench.py times. */
int uva_net_process_u8_device(uva_net* net, const void* d_in, int h, int w, size_t in_stride,
                              void* d_out, size_t out_stride, int tile_size, int border);

/* `count` frames of ONE geometry in one call, every d_in[i] / d_out[i] resident in this device's HBM: what a worker that
 * holds several decoded frames (the raw-video route, the frame pool, `-m a` in front of the 2x pass) hands over instead of
 * `count` calls of apply_model (upscale/upscale_processing.py:258-299) -- the reference has no such call, its unit is the frame.
 * The 1x HurrDeblur net runs up to eight of them per kernel launch (the launch's pipeline fill and the strips' warm-up rows
 * are then paid once per launch, not once per frame); every other net, and whatever does not fit, is processed frame by
 * frame.  The result bytes are exactly those of `count` uva_net_process_u8_device calls.  Asynchronous on the net's stream. */
int uva_net_process_u8_device_batch(uva_net* net, const void* const* d_in, void* const* d_out, int count, int h, int w,
                                    size_t in_stride, size_t out_stride, int tile_size, int border);

int uva_net_synchronize(uva_net* net);

/* Device-side ordering between two nets of one process (each net owns a stream): everything
 * `producer` has been asked to do so far completes before anything `net` is asked to do from now
 * on.  Used to chain the 1x HurrDeblur pass into the 2x pass without leaving the GPU (the reference
 * hops through an 8-bit PNG between them, upscale/upscale_processing.py:888-909; a u8 frame in HBM
 * carries exactly the same information). */
int uva_net_wait_for(uva_net* net, uva_net* producer);

/* ---- `-m n=K` film-grain denoise ------------------------------------------------------------ */

/* cv2.fastNlMeansDenoisingColored(cv2.UMat(img), None, K, K, 5, 9)   upscale/upscale_processing.py:350-354
 * (apply_denoise; K = 1..30 from `-m n=K`, :782-789): BGR -> 8-bit Lab (linear light), non-local means on L
 * with h_luma and on (a, b) with h_color (template 5, search 9), Lab -> BGR; on HIP device `device`.
 * in / out: host u8 HWC BGR [h][w][3].  OpenCV's integer weighting scheme and its 8-bit fixed-point Lab conversions
 * (RGB2Lab_b / Lab2RGBinteger: gamma tables, 2^12-scaled matrices, LabCbrtTab_b) are restated from the published 4.x
 * source (csrc/uva_denoise.hip.h; PARITY UNPINNED: no cv2 in this image to check a constant against). */
int uva_denoise_u8(int device, const uint8_t* in, int h, int w, size_t in_stride, uint8_t* out, size_t out_stride,
                   float h_luma, float h_color);
/* The same stage on a frame that is already in HBM (`-m n=K` in front of `-m a` and the 2x / 4x pass without leaving the
 * GPU: the reference's order, upscale/upscale_processing.py:880-886, 888-909, 913-920): d_in / d_out are device pointers, the
 * call returns once the work is queued on the stage's own stream.  `after` (may be null): everything that net has been
 * asked to do so far completes first; `before` (may be null): whatever that net is asked to do from now on waits for this
 * frame -- the two ends of uva_net_wait_for.  uva_denoise_synchronize waits for the stage's stream on `device`. */
int uva_denoise_u8_device(int device, const void* d_in, int h, int w, size_t in_stride, void* d_out, size_t out_stride,
                          float h_luma, float h_color, uva_net* after, uva_net* before);
int uva_denoise_synchronize(int device);
/* Test hook: one stage of the above on dense host arrays.  stage 0: BGR -> Lab [h][w][3]; 1: Lab -> BGR;
 * 2: non-local means on a 1-channel image; 3: on a 2-channel image (strength = h). */
int uva_debug_denoise_stage(int device, int stage, const uint8_t* in, int h, int w, float strength, uint8_t* out);

/* ---- introspection used by the parity tests and bench.py ------------------------------ */

/* Activation after convolution #conv_idx (+PReLU) of the last extract/process call with
 * tile_size <= 0, converted to host f32 planar [cout][h][w].  conv_idx in [0, nconv-2]. */
int uva_net_debug_read_activation(uva_net* net, int conv_idx, float* out_chw, int h, int w);

/* Per-kernel-kind timing with HIP events recorded on the net's stream around each launch.
 * kind: 0 head conv, 1 trunk conv (the dominant kernel), 2 tail conv.  Generic graphs (4x_Valar_v1): 1 = rdb4_kernel
 * (a residual dense block's first four convolutions, all planes of the frame), 2 = the block's 192 -> 64 convolution; 0 unused.
 * enable != 0 starts (and resets) collection; stats are valid after uva_net_synchronize. */
int uva_net_set_profiling(uva_net* net, int enable);
int uva_net_kernel_stats(uva_net* net, int kind, long long* launches, double* total_ms);

#ifdef UVA_INSTRUMENT   /* instrumented builds only (python -m upscale_video_amd.build --instrument) */
/* Debug: replays one trunk-layer launch on the last call's workspace with in-kernel cycle stamps
 * of workgroup 0 / wave 0: out[8*i + {0,1,2,3,4}] = tile i {start, k-loop done, barrier passed,
 * epilogue done, epilogue staging written} in s_memtime ticks (out holds 8*max_tiles values).
 * *tiles = tiles that workgroup processed.  ablate: 0 real kernel, 1 memory traffic only (no MFMA /
 * LDS reads), 2 compute only (L2-resident input, stores to a sink), 3 the u8 tail kernel instead, 4 memory
 * traffic and LDS fragment reads without MFMAs; bits 8.. = hundreds of timed launches instead of 10 (sustained,
 * power-limited state); *kernel_ms = mean launch time. */
int uva_net_debug_trunk_stamps(uva_net* net, unsigned long long* out, int max_tiles, int* tiles, int ablate,
                               float* kernel_ms);
/* Debug (generic graphs, UVA_RDB_STAMPS=1 in the environment): s_memtime stamps of rdb4_kernel's workgroup 0 in its last
 * launch, out[(step * 4 + wave) * 4 + {0: step start, 1: first part done, 2: at the barrier, 3: barrier passed}]. */
int uva_net_debug_rdb_stamps(uva_net* net, unsigned long long* out, int max_steps);
#endif

/* Test hook (host only, no device needed): the fp16 MFMA A-operand image convolution #conv_idx is
 * repacked into, [k-step][m-frag][lane][8].  *needed receives the element count.  conv_idx -1: the
 * last convolution as tail_kernel reads it (64-feature nets). */
int uva_net_debug_packed_weights(uva_net* net, int conv_idx, uint16_t* out, size_t out_halfs, size_t* needed);

/* Test hook (host only): the step lists trunk2_kernel (two fused trunk layers per launch) would walk for an
 * h x w frame cut into reference tiles (tile_size, border; <= 0: one plane) on `grid` workgroups.  Every
 * workgroup has `*stride` 32-byte entries of 8 words (csrc/uva_kernels.hip.h Trunk2Step), nsteps[b] of them
 * real.  plane_hw receives h, w, activation pitch and array offset (pixels) of up to max_planes planes.
 * Returns non-zero if steps_words (capacity in 32-bit words) is too small; *needed_words says how many. */
int uva_debug_trunk2_schedule(int h, int w, int tile_size, int border, int grid, uint32_t* steps_words,
                              size_t capacity_words, size_t* needed_words, int* nsteps, int* stride,
                              long long* plane_info, int max_planes, int* nplanes, long long* guard_bytes);

/* Test hook (host only): the same for trunkw_kernel (csrc/uva_wino.hip.h: the fused pair as 1-D Winograd F(2,3)), whose
 * segments start without the two input rows shared with the step above: entry layout in that file (TrunkwArgs). */
int uva_debug_trunkw_schedule(int h, int w, int tile_size, int border, int grid, uint32_t* steps_words,
                              size_t capacity_words, size_t* needed_words, int* nsteps, int* stride,
                              long long* plane_info, int max_planes, int* nplanes, long long* guard_bytes);

/* Test hook (host only, after load_param of a generic graph such as 4x_Valar_v1): what the executor planned --
 * info[0] dense chains sharing one array, [1] Concat layers, [2] of them free (every input already in place), [3] copying
 * their first input only, [4] 3x3 convolutions that take the LDS-tiled kernel, [5] channels of the widest shared array, [6] residual dense blocks
 * whose first four convolutions run as one launch (rdb4_kernel).  info: at least 8 ints. */
int uva_net_debug_generic_plan(const uva_net* net, int* info);

/* Test hook (host only): the work lists of the generic executor's persistent kernels for an h x w plane on `grid`
 * workgroups -- kind 0: rdb4_kernel (a residual dense block's first four convolutions, models/4x_Valar_v1.param:6-19),
 * kind 1 / 2: g_conv3_sw with 32 / 64-column strips (the 192 -> 64 and 64 -> 64 convolutions).  Every segment is 8 words
 * {c0, y_begin, y_end, own0, own1, plane, 0, 0}: the kernel computes columns c0.. of rows [y_begin, y_end) and writes
 * columns [own0, own1) of them; seg_begin (grid + 1 ints): workgroup g owns segments seg_begin[g] .. seg_begin[g+1]. */
int uva_debug_generic_segments(int kind, int h, int w, int grid, int32_t* segs_words, size_t capacity_words, size_t* needed_words,
                               int* seg_begin);
/* The same for a BATCH of planes sharing one launch (the reference tiles of a frame, upscale_processing.py:499-516):
 * dims = {h0, w0, h1, w1, ...}, nplanes <= 16; word 5 of a segment is the index of its plane. */
int uva_debug_generic_segments_planes(int kind, const int* dims, int nplanes, int grid, int32_t* segs_words, size_t capacity_words,
                                      size_t* needed_words, int* seg_begin);

/* Test hook (host only): how the generic executor groups the planes (reference tiles) of an h x w frame into the batches
 * that go through the graph together.  Four words per plane {h, w, batch, class}; planes of a batch have one class, at
 * most 16 members and -- unless a single plane exceeds it -- at most batch_pixels input pixels (<= 0: the default). */
int uva_debug_generic_batches(int h, int w, int tile_size, int border, long long batch_pixels, int32_t* words, size_t capacity_words,
                              size_t* needed_words);

/* Test hook (host only): the row lists sub10_kernel (the whole 24-feature 1x net, one launch) walks for an h x w
 * frame on `grid` workgroups.  Every workgroup has `*stride` 16-byte entries of 4 words {y, x0, emit, 0}
 * (csrc/uva_sub10.h Sub10Args), nrows[b] of them real: the kernel computes columns x0 .. x0+79 of row y and
 * writes columns x0+10 .. x0+69 of it when emit is set. */
int uva_debug_sub10_rows(int h, int w, int grid, uint32_t* rows_words, size_t capacity_words, size_t* needed_words,
                         int* nrows, int* stride);
/* ... for `frames` (1..8) frames of that geometry in one launch: the sequence runs over (frame, strip, row); the frame of an
 * entry is emit >> 8 (emit & 1 as above), the fourth word the distance to the nearest row of the segment that is written out. */
int uva_debug_sub10_rows_batch(int h, int w, int frames, int grid, uint32_t* rows_words, size_t capacity_words, size_t* needed_words,
                               int* nrows, int* stride);
/* The same for sub5_kernel (UVA_SUB5=1: that net as two launches of five layers, two pipelines -- a PAIR of 54-column strips --
 * per workgroup; csrc/uva_sub5.hip.h).  Entries {row y, column of computed column 0 of the pair's first strip, 1 = written
 * out, 0}; both launches walk the same lists. */
int uva_debug_sub5_rows(int h, int w, int grid, uint32_t* rows_words, size_t capacity_words, size_t* needed_words,
                        int* nrows, int* stride);

const char* uva_last_error(void);
int uva_abi_version(void);

#ifdef __cplusplus
}
#endif
#endif /* UVA_H */
