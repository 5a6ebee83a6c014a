"""MI355X mirror of the per-frame worker layer of davlee1972/upscale_video.

Same function names, argument meaning and error behaviour as the reference's
upscale/upscale_processing.py for the hot path only:

    get_frames        :27-37      logging_callback  :40-51     init_worker     :54-73
    apply_model       :258-299    process_model     :302-347   process_tile    :395-477
    upscale_image     :480-542    upscale_frames    :545-601
    apply_denoise     :350-361    process_denoise   :364-392   (`-m n=K`)

so that the reference's orchestrator (process_file, ffmpeg extract/merge; out of scope here) can
import these instead of its own.  Differences, all behind the same signatures:
  * `ncnn` is upscale_video_amd.ncnn (HIP kernels on MI355X), not ncnn_vulkan;
  * apply_model / upscale_image hand the whole u8 frame to one fused device call
    (Net.process_u8): normalise, every tile, *255 and the cv2 u8 conversion happen on the GPU and
    only u8 crosses PCIe.  The tile-by-tile float route of the reference is kept
    (FUSED_DEVICE_PATH = False) and is what the parity tests compare the fused route against;
  * PNG I/O goes through cv2 when present, Pillow otherwise (_imageio.py);
  * process_model / upscale_frames keep their signatures but, with PERSISTENT_WORKERS (default), hand
    the batch to a FramePool (frame_pool.py): the worker processes, their nets and workspaces survive
    from batch to batch instead of being rebuilt per call (reference :565-577, :923-948), and PNG
    decode / encode run beside the GPU instead of around it.  PERSISTENT_WORKERS = False is the
    reference-shaped route (a fresh spawn Pool + init_worker per call, `workers_used` offsets).
"""
import atexit
import logging
import math
import multiprocessing
import os
import sys

import numpy as np

from . import ncnn
from ._imageio import imread, imwrite

net = None
model_input_name = "input"
model_output_name = "output"
init_error = None      # why this worker has no net (init_worker never exits: a Pool respawns a dead worker forever)
PERSISTENT_WORKERS = True

TILE_SIZE = 960        # upscale_processing.py:489
TILE_BORDER = 10       # upscale_processing.py:409-427
FUSED_DEVICE_PATH = True


def get_frames(x):
    """'1,4-6' -> [1, 4, 5, 6]   (reference :27-37)"""
    frames = []
    for part in x.split(","):
        if "-" in part:
            lo, hi = (int(v) for v in part.split("-"))
            frames.extend(range(lo, hi + 1))
        else:
            frames.append(int(part))
    return frames


def logging_callback(log_list):
    """Parent-side sink for a worker's [[level, message], ...]; any error item ends the run
    (reference :40-51)."""
    failed = False
    for level, message in log_list:
        if level == "info":
            logging.info(message)
        elif level == "debug":
            logging.debug(message)
        elif level == "error":
            logging.error(message)
            failed = True
        if failed:
            sys.exit("Error - Exiting")


def _worker_slot(workers_used):
    ident = multiprocessing.current_process()._identity
    return (ident[0] - 1 - workers_used) if ident else 0


def workers_spawned_so_far():
    """How many pool identities this process has handed out: the `workers_used` a caller must pass to the
    reference-shaped route so that the next pool's workers map onto entries 0.. of its -g list (the reference
    threads the count by hand, :880-948; its own harness passes 0 because it starts from a fresh process,
    test_gpus.py:79-84).  Asking costs one identity, which is included in the answer."""
    return multiprocessing.get_context("spawn").Process()._identity[0]


def init_worker(gpus, workers_used, model_path, model_file, scale, model_input, model_output):
    """Pool initializer: position of this worker in the pool picks its entry of the -g list
    (duplicates allowed: '0,0,1' = two workers on GPU 0), then the net is built and loaded from
    models/<scale><model_file>.param|.bin  (reference :54-73)."""
    global net, model_input_name, model_output_name, init_error

    # A Pool replaces a worker that dies in its initializer, and the replacement (a higher pool
    # identity) dies the same way: pool.join() would never return.  So a worker that cannot be set up
    # stays alive without a net and every task it receives comes back as error items -- which is how
    # the reference reports a failed frame (:289-293) and what makes logging_callback end the run.
    net, init_error = None, None
    gpu = _worker_slot(workers_used)
    if gpu > len(gpus) - 1 or gpu < 0:
        init_error = "Unable to assign GPU to new worker."
    elif gpus[gpu] < 0:
        init_error = "GPU index %d: this build has no CPU path." % gpus[gpu]
    if init_error is None:
        try:
            candidate = ncnn.Net()
            candidate.opt.use_vulkan_compute = True
            candidate.set_vulkan_device(gpus[gpu])
            base = os.path.join(model_path, str(scale) + model_file)
            if candidate.load_param(base + ".param") or candidate.load_model(base + ".bin"):
                init_error = "Unable to load model %s: %s" % (base, getattr(candidate, "last_error", ""))
            else:
                net = candidate
        except Exception as e:  # noqa: BLE001 - libuva.so missing, no HIP device, ...
            init_error = "%s: %s" % (type(e).__name__, e)
    if init_error is not None:
        logging.error(init_error)
    model_input_name = model_input
    model_output_name = model_output


def _run_net(tile_bgr):
    """from_pixels(BGR) -> *1/255 -> extract -> np.array: f32 [3][h*s][w*s]  (reference :265-281)"""
    mat_in = ncnn.Mat.from_pixels(tile_bgr, ncnn.Mat.PixelType.PIXEL_BGR, tile_bgr.shape[1], tile_bgr.shape[0])
    mat_in.substract_mean_normalize([], [1 / 255.0, 1 / 255.0, 1 / 255.0])
    ex = net.create_extractor()
    ex.input(model_input_name, mat_in)
    ret, mat_out = ex.extract(model_output_name)
    if ret != 0:
        raise RuntimeError("extract failed")
    return np.array(mat_out)


def apply_model(input_file, output_file, remove):
    """1x whole-frame pass (the '-m a' HurrDeblur stage): PNG -> net -> PNG  (reference :258-299)."""
    logging_items = []
    if net is None:
        return [["error", "Model processing failed"], ["error", init_error or "worker has no net: init_worker was not run"]]
    img = imread(input_file)
    try:
        if img is None:
            raise RuntimeError("cannot read " + str(input_file))
        if FUSED_DEVICE_PATH:
            output = net.process_u8(img, tile_size=0)
        else:
            output = _run_net(img).transpose(1, 2, 0) * 255
        if output_file:
            imwrite(output_file, output)
    except Exception as e:  # noqa: BLE001 - the reference reports and carries on to the callback
        logging_items.append(["error", "Model processing failed"])
        logging_items.append(["error", e])
        ncnn.destroy_gpu_instance()
        return logging_items

    if remove:
        os.remove(input_file)
    logging_items.append(["info", "Processed Model: " + str(output_file)])
    return logging_items


def _pool(gpus, workers_used, model_path, model_file, scale, model_input, model_output):
    return multiprocessing.get_context("spawn").Pool(
        processes=len(gpus),
        initializer=init_worker,
        initargs=(gpus, workers_used, model_path, model_file, scale, model_input, model_output),
    )


class _Collector:
    """apply_async callback for the reference-shaped route.  The reference calls sys.exit from the
    callback (:47-51), i.e. inside the pool's result-handler thread, which ends that thread and not the
    program; here the failure is remembered and the main thread exits once the pool has been joined."""

    def __init__(self):
        self.failed = None

    def __call__(self, log_list):
        if self.failed is not None:
            return
        try:
            logging_callback(log_list)
        except SystemExit as e:
            self.failed = e

    def finish(self):
        if self.failed is not None:
            raise self.failed


def _check_gpus(gpus):
    """Parent-side validation before any worker exists (the reference would only find out inside a worker)."""
    if not gpus:
        logging.error("No GPUs given.")
        sys.exit("Error - Exiting")
    for g in gpus:
        if g < 0:
            logging.error("GPU index %d: this build has no CPU path." % g)
            sys.exit("Error - Exiting")


_frame_pools = {}
NET_FACTORY = None     # "module:function" override of frame_pool.load_reference_net (tests)


def get_frame_pool(gpus):
    """The persistent FramePool of this `-g` list, created on first use and kept until shutdown_workers()."""
    from . import frame_pool
    key = tuple(gpus)
    pool = _frame_pools.get(key)
    if pool is None or pool.closed or not all(p.is_alive() for p in pool.procs):
        if pool is not None:
            pool.close(timeout=1.0)
        kw = {"net_factory": NET_FACTORY} if NET_FACTORY else {}
        pool = frame_pool.FramePool(list(gpus), **kw)
        _frame_pools[key] = pool
    return pool


def shutdown_workers():
    """Ends every persistent worker (also registered with atexit)."""
    for pool in list(_frame_pools.values()):
        pool.close()
    _frame_pools.clear()


atexit.register(shutdown_workers)


def _run_persistent(gpus, tasks):
    from . import frame_pool
    pool = get_frame_pool(gpus)
    try:
        pool.run(tasks, callback=logging_callback)
    except frame_pool.WorkerDied as e:
        _frame_pools.pop(tuple(gpus), None)
        logging.error(e)
        sys.exit("Error - Exiting")
    except BaseException:
        # an error item (SystemExit from logging_callback), a dead worker or Ctrl-C: frames not yet
        # started keep their inputs, so a rerun picks up exactly what is missing
        pool.abort()
        _frame_pools.pop(tuple(gpus), None)
        raise


def process_model(frames_count, model_path, model_file, scale, model_input, model_output, input_file_tag,
                  output_file_tag, gpus, workers_used, remove=True):
    """Frame work queue for a 1x model: one spawned worker per -g entry, one task per existing
    '<n>.<input_tag>.png'  (reference :302-347).  No collective: frames are independent."""
    frames = range(1, frames_count + 1) if isinstance(frames_count, int) else frames_count
    _check_gpus(gpus)
    if PERSISTENT_WORKERS:
        tasks = []
        for frame in frames:
            src = "%s.%s.png" % (frame, input_file_tag)
            dst = "%s.%s.png" % (frame, output_file_tag)
            if os.path.exists(src):
                tasks.append(dict(src=src, dst=dst, model_path=model_path, model_file=model_file, scale=scale,
                                  tile_size=0, border=0, remove=remove,
                                  log_ok=[["info", "Processed Model: " + dst]],
                                  log_error=lambda e: [["error", "Model processing failed"], ["error", e]]))
        _run_persistent(gpus, tasks)
        return
    pool = _pool(gpus, workers_used, model_path, model_file, scale, model_input, model_output)
    collect = _Collector()
    for frame in frames:
        src = "%s.%s.png" % (frame, input_file_tag)
        dst = "%s.%s.png" % (frame, output_file_tag)
        if os.path.exists(src):
            pool.apply_async(apply_model, args=(src, dst, remove), callback=collect)
    pool.close()
    pool.join()
    collect.finish()


DENOISE_GPU = 0        # HIP device of a denoise worker (set per worker by _init_denoise_worker)


def denoise_u8(img_bgr, strength, device=None):
    """cv2.fastNlMeansDenoisingColored(img, None, K, K, 5, 9) on the MI355X (include/uva.h uva_denoise_u8)."""
    from . import _lib
    img = np.ascontiguousarray(img_bgr, dtype=np.uint8)
    if img.ndim != 3 or img.shape[2] != 3:
        raise ValueError("frame must be u8 [h][w][3]")
    h, w, _ = img.shape
    out = np.empty_like(img)
    _lib.check(_lib.load().uva_denoise_u8(DENOISE_GPU if device is None else int(device), img.ctypes.data, h, w, w * 3,
                                          out.ctypes.data, w * 3, float(strength), float(strength)))
    return out


def _imwrite_via_gpu(path, img):
    """cv2.imwrite for a frame that has just come off the GPU: PNG files are deflated there (ncnn.png_encode_u8,
    ~2 ms for a 1080p frame against 42 ms of zlib on a core); anything the encoder does not take goes through imwrite."""
    if str(path).lower().endswith(".png") and os.environ.get("UVA_GPU_PNG", "1") != "0":
        try:
            from . import ncnn
            data = ncnn.png_encode_u8(img, gpu=DENOISE_GPU)
        except ValueError:          # a frame shape the GPU encoder does not take
            data = None
        if data is not None:
            with open(path, "wb") as f:
                f.write(data)
            return True
    return imwrite(path, img)


def apply_denoise(input_file_name, output_file_name, denoise, remove):
    """One frame of the `-m n=K` film-grain pass: PNG -> non-local means -> PNG  (reference :350-361).  The
    reference lets an exception escape (the pool swallows it and the frame is silently missing); here a
    failure comes back as error items like every other worker function's."""
    try:
        img = imread(input_file_name)
        if img is None:
            raise RuntimeError("cannot read " + str(input_file_name))
        _imwrite_via_gpu(output_file_name, denoise_u8(img, denoise))
    except Exception as e:  # noqa: BLE001
        return [["error", "Denoise failed"], ["error", e]]
    if remove:
        os.remove(input_file_name)
    return [["info", "Processed Denoise: " + output_file_name]]


def _init_denoise_worker(gpus):
    global DENOISE_GPU
    slot = _worker_slot(0)
    DENOISE_GPU = gpus[slot % len(gpus)]


def process_denoise(frames_count, input_file_tag, denoise, remove=True, gpus=None, workers_per_gpu=4):
    """Frame work queue of the denoise pass (reference :364-392): one task per existing '<n>.<tag>.png', output
    '<n>.denoise.png'.  The reference spreads cv2's CPU / OpenCL code over a pool of os.cpu_count() processes;
    here the arithmetic runs on the GPUs of `gpus` (default: device 0) and the pool's processes only decode,
    encode and wait, `workers_per_gpu` of them per GPU.  Returns the number of pool processes like the
    reference (its caller adds it to workers_used, :885)."""
    frames = range(1, frames_count + 1) if isinstance(frames_count, int) else frames_count
    gpus = list(gpus) if gpus else [0]
    _check_gpus(gpus)
    nproc = len(gpus) * max(1, int(workers_per_gpu))
    pool = multiprocessing.get_context("spawn").Pool(nproc, initializer=_init_denoise_worker, initargs=(gpus,))
    collect = _Collector()
    for frame in frames:
        src = "%s.%s.png" % (frame, input_file_tag)
        dst = "%s.denoise.png" % frame
        if os.path.exists(src):
            pool.apply_async(apply_denoise, args=(src, dst, denoise, remove), callback=collect)
    pool.close()
    pool.join()
    collect.finish()
    return nproc


def tile_window(tile_size, y, x, height, width, border=TILE_BORDER):
    """Geometry of tile (y, x): core rectangle and the borders added on sides that are at least
    `border` px away from the image edge  (reference :398-427)."""
    y0, x0 = y * tile_size, x * tile_size
    y1, x1 = min(y0 + tile_size, height), min(x0 + tile_size, width)
    top = border if y0 >= border else 0
    bottom = border if y1 <= height - border else 0
    left = border if x0 >= border else 0
    right = border if x1 <= width - border else 0
    return (y0, y1, x0, x1), (top, bottom, left, right)


def process_tile(img, tile_size, scale, y, x, height, width, output, logging_items):
    """One tile through the float route: cut core+border, run the net, *255, paste the core
    (reference :395-477).  Returns -1 after reporting on failure."""
    (y0, y1, x0, x1), (top, bottom, left, right) = tile_window(tile_size, y, x, height, width)
    input_tile = img[y0 - top:y1 + bottom, x0 - left:x1 + right, :].copy()
    try:
        output_tile = _run_net(input_tile)
    except Exception as e:  # noqa: BLE001
        logging_items.append(["error", "Upscale failed"])
        logging_items.append(["error", e])
        logging.error(e)
        ncnn.destroy_gpu_instance()
        return -1
    output_tile = output_tile.transpose(1, 2, 0) * 255
    output[y0 * scale:y1 * scale, x0 * scale:x1 * scale, :] = output_tile[
        top * scale:(top + y1 - y0) * scale, left * scale:(left + x1 - x0) * scale, :]
    return 0


def upscale_image(input_file_name, output_file_name, scale, frame_batch, frame, end_frame, remove=True):
    """2x/4x of one frame with the reference's 960-px tiling  (reference :480-542)."""
    logging_items = []
    if net is None:
        return [["error", "Upscale failed"], ["error", init_error or "worker has no net: init_worker was not run"]]
    img = imread(input_file_name)
    if img is None:
        logging_items.append(["error", "Upscale failed"])
        logging_items.append(["error", "cannot read " + str(input_file_name)])
        return logging_items

    tile_size = TILE_SIZE
    height, width, batch = img.shape
    tiles_x = math.ceil(width / tile_size)
    tiles_y = math.ceil(height / tile_size)

    if FUSED_DEVICE_PATH:
        for idx in range(tiles_x * tiles_y):
            logging_items.append(["debug", f"Processing Tile: {idx + 1}/{tiles_x * tiles_y}"])
        try:
            output = net.process_u8(img, tile_size=tile_size, border=TILE_BORDER)
        except Exception as e:  # noqa: BLE001
            logging_items.append(["error", "Upscale failed"])
            logging_items.append(["error", e])
            logging.error(e)
            ncnn.destroy_gpu_instance()
            return logging_items
    else:
        output = np.zeros((height * scale, width * scale, batch))   # float64 canvas, :497
        for y in range(tiles_y):
            for x in range(tiles_x):
                logging_items.append(["debug", f"Processing Tile: {y * tiles_x + x + 1}/{tiles_x * tiles_y}"])
                if process_tile(img, tile_size, scale, y, x, height, width, output, logging_items) == -1:
                    return logging_items

    if output_file_name:
        imwrite(output_file_name, output)
    if remove:
        os.remove(input_file_name)

    logging_items.append(_progress_item(frame_batch, frame, end_frame, output_file_name))
    return logging_items


def _tile_items(shape, tile_size=None):
    """upscale_image's per-tile debug lines (:499-509) for a frame of `shape` = (height, width)"""
    if not shape:
        return []
    ts = tile_size or TILE_SIZE
    n = math.ceil(shape[1] / ts) * math.ceil(shape[0] / ts)
    return [["debug", "Processing Tile: %d/%d" % (i, n)] for i in range(1, n + 1)]


def _progress_item(frame_batch, frame, end_frame, output_file_name):
    """the reference's per-frame progress line (:524-540)"""
    if frame_batch:
        if isinstance(frame_batch, int):
            return ["info", "Upscaling Batch: %s : Upscaled %s/%s" % (frame_batch, frame, end_frame)]
        return ["info", "Upscaled " + str(output_file_name)]
    return ["info", "Upscaled %s/%s" % (frame, end_frame)]


def upscale_frames(frame_batch, start_frame, end_frame, input_file_tag, scale, gpus, workers_used, model_path,
                   model_file, model_input, model_output, remove=True):
    """Frame work queue for the 2x/4x pass of one batch  (reference :545-601)."""
    if frame_batch and isinstance(frame_batch, list):
        frames = frame_batch
    else:
        frames = range(start_frame, end_frame + 1)
    _check_gpus(gpus)
    if PERSISTENT_WORKERS:
        tasks = []
        for frame in frames:
            src = "%s.%s.png" % (frame, input_file_tag)
            dst = "%s.png" % frame
            if os.path.exists(src):
                # upscale_image's items (:507, :524-540): one "Processing Tile: i/n" debug line per reference tile -- the
                # worker reports the decoded frame's size with its result -- then the frame's progress line
                tasks.append(dict(src=src, dst=dst, model_path=model_path, model_file=model_file, scale=scale,
                                  tile_size=TILE_SIZE, border=TILE_BORDER, remove=remove,
                                  log_ok=lambda shape, item=_progress_item(frame_batch, frame, end_frame, dst): _tile_items(shape) + [item],
                                  log_error=lambda e: [["error", "Upscale failed"], ["error", e]]))
        _run_persistent(gpus, tasks)
        return
    pool = _pool(gpus, workers_used, model_path, model_file, scale, model_input, model_output)
    collect = _Collector()
    for frame in frames:
        src = "%s.%s.png" % (frame, input_file_tag)
        dst = "%s.png" % frame
        if os.path.exists(src):
            pool.apply_async(
                upscale_image,
                args=(src, dst, scale, frame_batch, frame, end_frame, remove),
                callback=collect,
            )
    pool.close()
    pool.join()
    collect.finish()
