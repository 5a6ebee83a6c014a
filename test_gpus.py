"""test_gpus.py -- device listing and timing harness for the MI355X engine.

Counterpart of the reference's only benchmark harness (test_gpus.py:15-127) with the same flags:
  -g "0,1,1,2"  one spawned worker per entry (duplicates = several workers on one GPU)
  -s 2|4        scale
  -r N          number of upscale_image calls spread over the pool
Without -g it only lists the HIP devices.  It reports what the reference reports -- the seconds of
every call and the seconds of the whole run -- plus frames/s.  The reference's sample.png is a missing
blob upstream (.MISSING_LARGE_BLOBS); without -i a seeded synthetic 1920x1080 frame is written to a
temporary PNG instead.
"""
import argparse
import logging
import multiprocessing as mp
import os
import sys
import tempfile
import time

from upscale_video_amd import ncnn
from upscale_video_amd.synth import synthetic_frame
from upscale_video_amd.upscale_processing import init_worker, logging_callback, upscale_image, workers_spawned_so_far

RULE = "=" * 36
DEVICE_KINDS = ("Discrete", "Integrated", "Virtual", "CPU")     # ncnn's gpu_info.type() enumeration


def timed_upscale(png, scale, gpus, workers_used=0):
    """Worker task: one upscale_image call on this worker's net, timed; returns log items."""
    me = mp.current_process()._identity
    slot = me[0] - 1 - workers_used if me else 0
    t0 = time.perf_counter()
    log = [["info", "Testing GPU: %s" % gpus[slot]]]
    log.extend(upscale_image(png, None, scale, None, 1, 1, remove=False))
    dt = time.perf_counter() - t0
    log.append(["info", "%s seconds to upscale %s" % (dt, os.path.basename(png))])
    log.append(["debug", "%s %d %s %.6f" % (WORKER_TIME_TAG, slot, gpus[slot], dt)])      # for the parent's per-worker table
    return log


WORKER_TIME_TAG = "worker-time"


def per_worker_table(times, elapsed):
    """{(slot, gpu): [seconds per call]} -> log lines: one per worker, slowest last but one, then the spread.  The reference
    prints seconds per call and the total (test_gpus.py:79-112); with `-g 0,..,7` the question after a run is WHICH worker
    was slow, so the calls are also shown per worker."""
    lines = ["per worker (calls, mean seconds per call, frames/s of the worker while it ran):"]
    rates = []
    for (slot, gpu), ts in sorted(times.items()):
        rate = len(ts) / sum(ts)
        rates.append(rate)
        lines.append("  worker %d on GPU %s: %d calls, %.4f s per call, %.2f frames/s" % (slot, gpu, len(ts), sum(ts) / len(ts), rate))
    if rates:
        lines.append("  slowest / fastest worker: %.2f / %.2f frames/s; sum of the workers' rates %.2f, the pool's %.2f"
                     % (min(rates), max(rates), sum(rates), sum(len(t) for t in times.values()) / elapsed))
    return lines


def list_devices():
    n = ncnn.get_gpu_count()
    for line in ("Searching for HIP (MI355X) GPUs", RULE, "GPU count: %d" % n, RULE,
                 "Default GPU: %d" % ncnn.get_default_gpu_index(), RULE):
        logging.info(line)
    for idx in range(n):
        info = ncnn.get_gpu_info(idx)
        logging.info("GPU %d: %s / %s", idx, DEVICE_KINDS[info.type()], info.device_name())
    return n


def time_pool(gpu_list, scale, runs, png):
    models = os.path.join(os.path.dirname(os.path.realpath(__file__)), "models")
    used = workers_spawned_so_far()        # 0 + 1 in a fresh process; later pools of a long-lived caller continue the count
    workers = mp.get_context("spawn").Pool(
        len(gpu_list), init_worker, (gpu_list, used, models, "x_Compact_Pretrain", scale, "input", "output"))
    for line in ("", "Starting test runs", RULE,
                 "(what is timed, as in the reference's harness: every run's PNG decode and worker start-up -- the pool's spawn "
                 "and model load fall inside the total; a host-side figure, not the GPUs' rate, which is bench.py's)", RULE):
        logging.info(line)
    times = {}

    def collect(log_list):
        for level, message in log_list:
            if level == "debug" and message.startswith(WORKER_TIME_TAG + " "):
                _, slot, gpu, dt = message.split()
                times.setdefault((int(slot), gpu), []).append(float(dt))
        logging_callback(log_list)

    t0 = time.perf_counter()
    for _ in range(runs):
        workers.apply_async(timed_upscale, (png, scale, gpu_list, used), callback=collect)
    workers.close()
    workers.join()
    elapsed = time.perf_counter() - t0
    logging.info(RULE)
    for line in per_worker_table(times, elapsed):
        logging.info(line)
    logging.info(RULE)
    logging.info("%s seconds total to run tests.", elapsed)
    logging.info("%.3f frames/s file-to-HBM-to-host (PNG decode included, no encode)", runs / elapsed)


def run_tests(gpus=None, scale=2, runs=10, image=None, size="1920x1080"):
    logging.basicConfig(level=logging.INFO, stream=sys.stdout, datefmt="%Y-%m-%d %H:%M:%S",
                        format="[%(asctime)s] [%(levelname)s] %(message)s")
    list_devices()
    if gpus is None:
        return
    gpu_list = [int(tok) for tok in gpus.split(",")] if gpus else [0]
    with tempfile.TemporaryDirectory(prefix="test_gpus_") as scratch:
        if image is None:
            from upscale_video_amd._imageio import imwrite
            image = os.path.join(scratch, "sample.png")
            w, h = (int(v) for v in size.lower().split("x"))
            imwrite(image, synthetic_frame(h, w))
            logging.info("sample.png: a synthetic %dx%d frame", w, h)
        time_pool(gpu_list, scale, runs, image)


if __name__ == "__main__":
    cli = argparse.ArgumentParser(description="List the HIP GPUs; with -g, time upscale_image on them")
    cli.add_argument("-g", "--gpus", help="worker list, one worker per entry, e.g. 0,1,1,2")
    cli.add_argument("-s", "--scale", type=int, default=2, choices=(2, 4), help="2 (default) or 4")
    cli.add_argument("-r", "--runs", type=int, default=10, help="number of timed calls (default 10)")
    cli.add_argument("-i", "--image", help="PNG to upscale (default: a synthetic frame of --size)")
    cli.add_argument("--size", default="1920x1080", help="WxH of the synthetic frame (BASELINE config 5: 3840x2160 with -g 0,1,2,3,4,5,6,7)")
    opts = cli.parse_args()
    run_tests(opts.gpus, opts.scale, opts.runs, opts.image, opts.size)
