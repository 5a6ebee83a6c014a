"""test_gpus.py -- the reference's only benchmark harness (test_gpus.py:15-127), on the MI355X engine.

Same flags: -g "0,1,1,2" (one spawned worker per entry), -s 2|4, -r runs.  Lists the HIP devices,
then times `runs` calls of upscale_image on one frame across the pool and prints the seconds of
every call and the total, exactly the two figures the reference prints (fps = runs / total).
sample.png is a missing blob in the reference (.MISSING_LARGE_BLOBS); without -i a seeded
synthetic 1920x1080 frame is written to a temp file instead.
"""
import argparse
import logging
import multiprocessing
import os
import sys
import tempfile
import time

import numpy as np

from upscale_video_amd import ncnn
from upscale_video_amd.upscale_processing import init_worker, logging_callback, upscale_image


def upscale_images(input_file_name, output_file_name, scale, gpus):
    ident = multiprocessing.current_process()._identity
    i = (int(ident[0]) - 1) if ident else 0
    start = time.time()
    items = [["info", "Testing GPU: " + str(gpus[i])]]
    items += upscale_image(input_file_name, output_file_name, scale, None, 1, 1, remove=False)
    items.append(["info", str(time.time() - start) + " seconds to upscale " + os.path.basename(input_file_name)])
    return items


def synthetic_png(path, h=1080, w=1920, seed=20260928):
    from upscale_video_amd._imageio import imwrite
    rng = np.random.default_rng(seed)
    y, x = np.mgrid[0:h, 0:w].astype(np.float64)
    img = np.stack([127 + 100 * np.sin(x / 17 + c) * np.cos(y / 23) + rng.normal(0, 4, (h, w)) for c in range(3)], -1)
    imwrite(path, np.clip(np.rint(img), 0, 255).astype(np.uint8))


def run_tests(gpus=None, scale=2, runs=10, image=None):
    logging.basicConfig(level=logging.INFO, format="[%(asctime)s] [%(levelname)s] %(message)s",
                        datefmt="%Y-%m-%d %H:%M:%S", stream=sys.stdout)
    gpu_count = ncnn.get_gpu_count()
    logging.info("Searching for HIP (MI355X) GPUs")
    logging.info("====================================")
    logging.info("GPU count: " + str(gpu_count))
    logging.info("====================================")
    logging.info("Default GPU: " + str(ncnn.get_default_gpu_index()))
    logging.info("====================================")
    gpu_types = ["Discrete", "Integrated", "Virtual", "CPU"]
    for i in range(gpu_count):
        info = ncnn.get_gpu_info(i)
        logging.info("GPU %d: %s / %s" % (i, gpu_types[info.type()], info.device_name()))

    if gpus is None:
        return
    gpus = [int(g) for g in gpus.split(",")] if gpus else [0]
    here = os.path.dirname(os.path.realpath(__file__))
    model_path = os.path.join(here, "models")
    tmp = None
    if image is None:
        tmp = tempfile.mkdtemp(prefix="test_gpus_")
        image = os.path.join(tmp, "sample.png")
        synthetic_png(image)

    pool = multiprocessing.get_context("spawn").Pool(
        processes=len(gpus), initializer=init_worker,
        initargs=(gpus, 0, model_path, "x_Compact_Pretrain", scale, "input", "output"))
    logging.info("")
    logging.info("Starting test runs")
    logging.info("====================================")
    start = time.time()
    for _ in range(runs):
        pool.apply_async(upscale_images, args=(image, None, scale, gpus), callback=logging_callback)
    pool.close()
    pool.join()
    total = time.time() - start
    logging.info("====================================")
    logging.info(str(total) + " seconds total to run tests.")
    logging.info("%.3f frames/s file-to-HBM-to-host (PNG decode included, no encode)" % (runs / total))
    if tmp:
        os.remove(image)
        os.rmdir(tmp)


if __name__ == "__main__":
    parser = argparse.ArgumentParser(description="Test GPU - List GPUs")
    parser.add_argument("-g", "--gpus", help="Optional gpus to test. Example 0,1,1,2. Default is 0.")
    parser.add_argument("-s", "--scale", type=int, default=2, help="Scale 2 or 4. Default is 2.")
    parser.add_argument("-r", "--runs", type=int, default=10, help="Number of tests")
    parser.add_argument("-i", "--image", help="PNG to upscale (default: synthetic 1920x1080 frame)")
    args = parser.parse_args()
    run_tests(args.gpus, args.scale, args.runs, args.image)
