"""MI355X mirror of the per-frame worker layer of davlee1972/upscale_video.

Same function names, argument meaning and error behaviour as the reference's
upscale/upscale_processing.py for the hot path only:

    get_frames        :27-37      logging_callback  :40-51     init_worker     :54-73
    apply_model       :258-299    process_model     :302-347   process_tile    :395-477
    upscale_image     :480-542    upscale_frames    :545-601

so that the reference's orchestrator (process_file, ffmpeg extract/merge; out of scope here) can
import these instead of its own.  Differences, all behind the same signatures:
  * `ncnn` is upscale_video_amd.ncnn (HIP kernels on MI355X), not ncnn_vulkan;
  * apply_model / upscale_image hand the whole u8 frame to one fused device call
    (Net.process_u8): normalise, every tile, *255 and the cv2 u8 conversion happen on the GPU and
    only u8 crosses PCIe.  The tile-by-tile float route of the reference is kept
    (FUSED_DEVICE_PATH = False) and is what the parity tests compare the fused route against;
  * PNG I/O goes through cv2 when present, Pillow otherwise (_imageio.py).
"""
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


def init_worker(gpus, workers_used, model_path, model_file, scale, model_input, model_output):
    """Pool initializer: position of this worker in the pool picks its entry of the -g list
    (duplicates allowed: '0,0,1' = two workers on GPU 0), then the net is built and loaded from
    models/<scale><model_file>.param|.bin  (reference :54-73)."""
    global net, model_input_name, model_output_name

    gpu = _worker_slot(workers_used)
    if gpu > len(gpus) - 1:
        logging.error("Unable to assign GPU to new worker.")
        sys.exit("Error - Exiting")
    if gpus[gpu] < 0:
        logging.error("GPU index %d: this build has no CPU path." % gpus[gpu])
        sys.exit("Error - Exiting")

    net = ncnn.Net()
    net.opt.use_vulkan_compute = True
    net.set_vulkan_device(gpus[gpu])

    base = os.path.join(model_path, str(scale) + model_file)
    if net.load_param(base + ".param") or net.load_model(base + ".bin"):
        logging.error("Unable to load model %s: %s" % (base, getattr(net, "last_error", "")))
        sys.exit("Error - Exiting")
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


def process_model(frames_count, model_path, model_file, scale, model_input, model_output, input_file_tag,
                  output_file_tag, gpus, workers_used, remove=True):
    """Frame work queue for a 1x model: one spawned worker per -g entry, one task per existing
    '<n>.<input_tag>.png'  (reference :302-347).  No collective: frames are independent."""
    frames = range(1, frames_count + 1) if isinstance(frames_count, int) else frames_count
    pool = _pool(gpus, workers_used, model_path, model_file, scale, model_input, model_output)
    for frame in frames:
        src = "%s.%s.png" % (frame, input_file_tag)
        dst = "%s.%s.png" % (frame, output_file_tag)
        if os.path.exists(src):
            pool.apply_async(apply_model, args=(src, dst, remove), callback=logging_callback)
    pool.close()
    pool.join()


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

    if frame_batch:
        if isinstance(frame_batch, int):
            logging_items.append(["info", "Upscaling Batch: %s : Upscaled %s/%s" % (frame_batch, frame, end_frame)])
        else:
            logging_items.append(["info", "Upscaled " + str(output_file_name)])
    else:
        logging_items.append(["info", "Upscaled %s/%s" % (frame, end_frame)])
    return logging_items


def upscale_frames(frame_batch, start_frame, end_frame, input_file_tag, scale, gpus, workers_used, model_path,
                   model_file, model_input, model_output, remove=True):
    """Frame work queue for the 2x/4x pass of one batch  (reference :545-601)."""
    if frame_batch and isinstance(frame_batch, list):
        frames = frame_batch
    else:
        frames = range(start_frame, end_frame + 1)
    pool = _pool(gpus, workers_used, model_path, model_file, scale, model_input, model_output)
    for frame in frames:
        src = "%s.%s.png" % (frame, input_file_tag)
        dst = "%s.png" % frame
        if os.path.exists(src):
            pool.apply_async(
                upscale_image,
                args=(src, dst, scale, frame_batch, frame, end_frame, remove),
                callback=logging_callback,
            )
    pool.close()
    pool.join()
