"""frame_pool.py -- persistent per-GPU frame workers.

The reference builds a fresh `multiprocessing.Pool` -- and with it a fresh ncnn net, pipeline cache
and weight upload per worker -- for every 10-minute batch (upscale/upscale_processing.py:565-577,
:923-948), and each of its workers does imread -> net -> imwrite strictly one after the other
(:487, :505-516, :519), so the PNG codecs sit on the GPU's critical path.  A FramePool keeps the
reference's partition of the work (one spawned worker process per `-g` entry, duplicates allowed;
frames are independent units pulled from one shared queue; no collective) and its resume semantics
(a task exists only for an input file that exists, :339/:585; the input is removed only after the
output has been written, :295-296/:521-522), but

  * the workers live as long as the pool: nets, packed weights, activation workspaces and the
    page-locked result ring are built once per (worker, model) and reused by every later batch;
  * inside a worker, PNG decode, the GPU and PNG encode are three overlapped stages: a pool of
    decode threads feeds the GPU thread (pipelined `submit_u8_png` / `collect_u8`, three frames in
    flight): the result frame never leaves HBM, a kernel behind the net deflates it (csrc/uva_png.hip.h)
    and only the compressed blocks cross PCIe; a pool of encode threads frames them as a PNG file
    (zlib header, Adler-32, chunk CRC) and writes it.  UVA_GPU_PNG=0 keeps the frame route instead
    (`submit_u8`, zlib on the host's cores: ~180 ms of a core per 3840x2160 frame);
  * every finished frame is reported to the caller's thread as the reference's log-item list, so
    `logging_callback`'s "any error item ends the run" (:40-51) happens in the main thread.

A worker that dies (or a run that is interrupted) loses only the frames it had in flight: their
inputs still exist, so the next run redoes exactly those.
"""
import importlib
import multiprocessing
import os
import queue
import threading
import time
from concurrent.futures import ThreadPoolExecutor

from ._imageio import imread, imwrite

PIPE_DEPTH = 3            # include/uva.h: frames in flight per net on the pipelined route
DEFAULT_DECODE_THREADS = int(os.environ.get("UVA_DECODE_THREADS", "0"))     # 0: the host's cores shared out over the workers
DEFAULT_ENCODE_THREADS = int(os.environ.get("UVA_ENCODE_THREADS", "0"))     # 0: likewise
GPU_PNG = os.environ.get("UVA_GPU_PNG", "1") != "0"        # imwrite's deflate work on the GPU (ncnn.Net.submit_u8_png)


def default_decode_threads(n_workers):
    """With the result frames deflated on the GPU, PNG decode of the inputs (~23 ms of a core per 1080p frame) is the
    host's main job: every worker gets its share of the cores, between 4 and 16 threads (measured on a 16-core
    quota: -g 0 best with 16, -g 0,0 with 8, profiles/r02_g_png_gpu_route.txt).  On the zlib route: 4."""
    if not GPU_PNG:
        return 4
    return max(4, min(16, usable_cpus() // max(1, n_workers)))


def default_encode_threads(n_workers, decode_threads):
    """zlib route: PNG encode of a 4K result is ~100x the GPU time of the frame, so every worker gets its share of the
    host's cores (minus its decode threads and its GPU thread), between 4 and 48 threads.  GPU route: the encode
    threads only frame the file (concatenate, CRC-32: ~5 ms) and write it -- 2 to 4."""
    cores = usable_cpus()
    if GPU_PNG:
        return max(2, min(4, cores // max(1, 2 * n_workers)))
    return max(4, min(48, cores // max(1, n_workers) - decode_threads - 1))


def usable_cpus():
    """CPUs this process may actually use: the affinity mask capped by the container's CFS quota (cgroup v2
    cpu.max / v1 cpu.cfs_quota_us) -- a box that shows 256 CPUs may grant 16."""
    n = len(os.sched_getaffinity(0)) if hasattr(os, "sched_getaffinity") else (os.cpu_count() or 8)
    try:
        quota, period = open("/sys/fs/cgroup/cpu.max").read().split()[:2]
        if quota != "max":
            n = min(n, max(1, int(int(quota) / int(period))))
    except Exception:  # noqa: BLE001
        try:
            q = int(open("/sys/fs/cgroup/cpu/cpu.cfs_quota_us").read())
            p = int(open("/sys/fs/cgroup/cpu/cpu.cfs_period_us").read())
            if q > 0:
                n = min(n, max(1, q // p))
        except Exception:  # noqa: BLE001
            pass
    return n
DEFAULT_NET_FACTORY = "upscale_video_amd.frame_pool:load_reference_net"


def load_reference_net(model_path, model_file, scale, gpu):
    """models/<scale><model_file>.param|.bin on HIP device `gpu`, as init_worker builds it (:65-71)."""
    from . import ncnn
    if gpu < 0:
        raise RuntimeError("GPU index %d: this build has no CPU path." % gpu)
    net = ncnn.Net()
    net.opt.use_vulkan_compute = True
    net.set_vulkan_device(gpu)
    base = os.path.join(model_path, str(scale) + model_file)
    if net.load_param(base + ".param") or net.load_model(base + ".bin"):
        raise RuntimeError("Unable to load model %s: %s" % (base, getattr(net, "last_error", "")))
    return net


def _resolve(path):
    mod, _, name = path.partition(":")
    return getattr(importlib.import_module(mod), name)


class _Worker:
    """Body of one worker process: decode threads -> GPU thread -> encode threads."""

    def __init__(self, gpu, slot, task_q, result_q, factory, decode_threads, encode_threads):
        self.gpu, self.slot = gpu, slot
        self.task_q, self.result_q = task_q, result_q
        self.factory = _resolve(factory)
        self.nets = {}
        self.rings = {}
        self.decoders = ThreadPoolExecutor(decode_threads, thread_name_prefix="decode")
        self.encoders = ThreadPoolExecutor(encode_threads, thread_name_prefix="encode")
        # frames resident in this worker (decoding, decoded, on the GPU, encoding): bounds its memory
        self.resident = threading.Semaphore(decode_threads + PIPE_DEPTH + encode_threads)
        self.nbuf = PIPE_DEPTH + encode_threads       # result buffers: frames on the GPU or being encoded
        self.ready = queue.Queue()      # decoded frames, in completion order
        self.stop = False
        self.gpu_png = GPU_PNG

    # ---- stage 1: pull tasks, decode --------------------------------------------------------
    def puller(self):
        while True:
            self.resident.acquire()
            task = self.task_q.get()
            if task is None:
                self.resident.release()
                self.ready.put(None)
                return
            self.decoders.submit(self.decode, task)

    def decode(self, task):
        try:
            img = imread(task["src"])
            if img is None:
                raise RuntimeError("cannot read " + str(task["src"]))
            self.ready.put((task, img, None))
        except Exception as e:  # noqa: BLE001
            self.ready.put((task, None, e))

    # ---- stage 2: the GPU thread (the only one that touches HIP) ---------------------------------
    def net_for(self, task):
        key = (task["model_path"], task["model_file"], task["scale"])
        if key not in self.nets:
            self.nets[key] = self.factory(task["model_path"], task["model_file"], task["scale"], self.gpu)
        return key, self.nets[key]

    def out_buffer(self, key, shape):
        """Result ring of one (model, frame size): page-locked when the engine offers it.  A buffer is
        reused only after its encode has finished; with every buffer busy the GPU thread waits here for
        the encoders (back-pressure)."""
        ring = self.rings.get((key, shape))
        if ring is None:
            try:
                from .ncnn import pinned_empty as alloc
                alloc((1,))
            except Exception:  # noqa: BLE001 - a stand-in net without the engine
                import numpy as np
                alloc = lambda s: np.empty(s, np.uint8)   # noqa: E731
            ring = {"bufs": [alloc(shape) for _ in range(self.nbuf)], "free": queue.Queue()}
            for i in range(self.nbuf):
                ring["free"].put(i)
            self.rings[(key, shape)] = ring
        return ring, ring["free"].get()

    def png_ring(self, key, h, w):
        """Ring of page-locked PNG workspaces for h x w result frames (ncnn.PngWorkspace), or None when the GPU encoder
        does not take such frames."""
        rk = (key, "png", h, w)
        if rk not in self.rings:
            from .ncnn import PngWorkspace
            try:
                ring = {"bufs": [PngWorkspace(h, w) for _ in range(self.nbuf)], "free": queue.Queue()}
            except ValueError:
                ring = None
            else:
                for i in range(self.nbuf):
                    ring["free"].put(i)
            self.rings[rk] = ring
        return self.rings[rk]

    def gpu_loop(self):
        inflight = []   # (task, ticket, net, ring, idx)
        eof = False
        while not eof or inflight:
            item = None
            if not eof and len(inflight) < PIPE_DEPTH:
                try:
                    item = self.ready.get(block=not inflight)
                except queue.Empty:
                    item = False
                if item is None:
                    eof = True
                    continue
            if item:
                task, img, err = item
                if err is not None:
                    self.finish(task, err)
                    continue
                task["_shape"] = (int(img.shape[0]), int(img.shape[1]))     # for the caller's per-tile log lines
                try:
                    key, net = self.net_for(task)
                    s = net.scale
                    tile = task["tile_size"]
                    ring = None
                    if self.gpu_png and task["dst"] and str(task["dst"]).lower().endswith(".png") and hasattr(net, "submit_u8_png"):
                        ring = self.png_ring(key, img.shape[0] * s, img.shape[1] * s)
                    if ring is not None:
                        idx = ring["free"].get()
                    else:
                        ring, idx = self.out_buffer(key, (img.shape[0] * s, img.shape[1] * s, 3))
                    try:
                        if hasattr(ring["bufs"][idx], "file_bytes"):
                            ticket = net.submit_u8_png(img, workspace=ring["bufs"][idx], tile_size=tile, border=task["border"] if tile else 0)
                        else:
                            ticket = net.submit_u8(img, out=ring["bufs"][idx], tile_size=tile, border=task["border"] if tile else 0)
                    except BaseException:
                        ring["free"].put(idx)           # the buffer goes back: a failed frame must not shrink the ring
                        raise
                    inflight.append((task, ticket, net, ring, idx))
                except Exception as e:  # noqa: BLE001
                    self.gpu_failed(task, e)
                continue
            # nothing new to submit (pipe full, or no decoded frame waiting): retire the oldest frame
            task, ticket, net, ring, idx = inflight.pop(0)
            try:
                out = net.collect_u8(ticket)
                self.encoders.submit(self.encode, task, out, ring, idx)
            except Exception as e:  # noqa: BLE001
                ring["free"].put(idx)
                self.gpu_failed(task, e)

    def gpu_failed(self, task, e):
        try:   # the reference tears the GPU instance down on a failed extract (:292, :458)
            from . import ncnn
            ncnn.destroy_gpu_instance()
        except Exception:  # noqa: BLE001
            pass
        self.nets.clear()
        self.finish(task, e)

    # ---- stage 3: encode, delete-after-write, report --------------------------------------------
    def encode(self, task, out, ring, idx):
        err = None
        try:
            if task["dst"]:
                if hasattr(out, "file_bytes"):      # a PNG workspace the GPU filled: frame it and write it
                    with open(task["dst"], "wb") as f:
                        f.write(out.file_bytes())
                else:
                    imwrite(task["dst"], out)
            if task["remove"]:
                os.remove(task["src"])
        except Exception as e:  # noqa: BLE001
            err = e
        ring["free"].put(idx)
        self.finish(task, err)

    def finish(self, task, err):
        self.resident.release()
        self.result_q.put((task["id"], self.slot, None if err is None else "%s: %s" % (type(err).__name__, err), task.get("_shape")))

    def run(self):
        t = threading.Thread(target=self.puller, daemon=True)
        t.start()
        self.gpu_loop()
        self.encoders.shutdown(wait=True)
        self.decoders.shutdown(wait=True)


def _worker_main(gpu, slot, task_q, result_q, factory, decode_threads, encode_threads, affinity):
    if affinity:
        try:
            os.sched_setaffinity(0, affinity)
        except OSError:
            pass
    try:
        _Worker(gpu, slot, task_q, result_q, factory, decode_threads, encode_threads).run()
    except BaseException as e:  # noqa: BLE001 - report, then die: the parent sees both
        result_q.put((None, slot, "worker %d (GPU %d) failed: %s: %s" % (slot, gpu, type(e).__name__, e)))
        raise


class WorkerDied(RuntimeError):
    pass


class FramePool:
    """One persistent worker process per entry of `gpus` (duplicates = several workers on one GPU)."""

    def __init__(self, gpus, net_factory=DEFAULT_NET_FACTORY, decode_threads=DEFAULT_DECODE_THREADS,
                 encode_threads=DEFAULT_ENCODE_THREADS, numa_affinity=True):
        if not gpus:
            raise ValueError("FramePool needs at least one -g entry")
        self.gpus = list(gpus)
        if not decode_threads:
            decode_threads = default_decode_threads(len(self.gpus))
        if not encode_threads:
            encode_threads = default_encode_threads(len(self.gpus), decode_threads)
        self.decode_threads, self.encode_threads = decode_threads, encode_threads
        ctx = multiprocessing.get_context("spawn")     # the reference's start method (:321, :565)
        self.task_q = ctx.Queue()
        self.result_q = ctx.Queue()
        self.procs = []
        self.next_id = 0
        self.closed = False
        for slot, gpu in enumerate(self.gpus):
            aff = gpu_cpu_affinity(gpu) if numa_affinity else None
            p = ctx.Process(target=_worker_main, daemon=True,
                            args=(gpu, slot, self.task_q, self.result_q, net_factory, decode_threads, encode_threads, aff))
            p.start()
            self.procs.append(p)

    def run(self, tasks, callback=None, poll=0.2):
        """Queues `tasks` (dicts: src, dst, model_path, model_file, scale, tile_size, border, remove,
        log_ok -- a list of log items or a function of the decoded frame's (height, width) returning one --, log_error)
        and blocks until every one of them has been reported.  `callback(log_items)` runs in the
        caller's thread once per frame, errors included.  Raises WorkerDied if a worker process ends
        while frames are outstanding (the frames it held are left for the next run)."""
        if self.closed:
            raise RuntimeError("FramePool is closed")
        pending = {}
        for t in tasks:
            tid = self.next_id
            self.next_id += 1
            pending[tid] = t
            wire = {k: v for k, v in t.items() if k not in ("log_ok", "log_error")}   # callables stay here
            # The workers keep the directory they were spawned in; the caller does not (the reference's orchestrator
            # changes into every file's temp dir, upscale/upscale_processing.py:842, back at :971, and hands out names
            # relative to it): paths are made absolute HERE, against the caller's directory at the time of the call.
            for key in ("src", "dst", "model_path"):
                if wire.get(key):
                    wire[key] = os.path.abspath(wire[key])
            wire["id"] = tid
            self.task_q.put(wire)
        done = 0
        while pending:
            try:
                tid, slot, err, *rest = self.result_q.get(timeout=poll)
            except queue.Empty:
                dead = [i for i, p in enumerate(self.procs) if not p.is_alive()]
                if dead:
                    self.abort()
                    raise WorkerDied("worker(s) %s ended with %d frame(s) outstanding" % (dead, len(pending)))
                continue
            if tid is None:
                self.abort()
                raise WorkerDied(err)
            task = pending.pop(tid, None)
            if task is None:
                continue        # a frame of an aborted earlier run
            done += 1
            items = task["log_error"](err) if err is not None else task["log_ok"]
            if callable(items):     # log_ok(shape): the frame's (height, width) as the worker decoded it (None for stand-ins)
                items = items(rest[0] if rest else None)
            if callback is not None:
                callback(items)
        return done

    def abort(self):
        """Drop whatever is still queued (frames not started keep their inputs) and stop the workers."""
        try:
            while True:
                self.task_q.get_nowait()
        except queue.Empty:
            pass
        self.close(timeout=1.0)

    def close(self, timeout=30.0):
        if self.closed:
            return
        self.closed = True
        for _ in self.procs:
            self.task_q.put(None)
        t_end = time.time() + timeout
        for p in self.procs:
            p.join(max(0.0, t_end - time.time()))
            if p.is_alive():
                p.terminate()
                p.join(1.0)

    def __enter__(self):
        return self

    def __exit__(self, *exc):
        self.close()


def gpu_cpu_affinity(gpu):
    """CPUs of the NUMA node the GPU's PCIe function hangs off (sysfs), or None when that cannot be
    told: feeder threads and pinned staging buffers then sit next to the GPU they serve."""
    try:
        from . import _lib
        import ctypes
        buf = ctypes.create_string_buffer(64)
        if _lib.load().uva_get_gpu_pci_bus_id(int(gpu), buf, 64) != 0:
            return None
        bdf = buf.value.decode().lower()
        with open("/sys/bus/pci/devices/%s/numa_node" % bdf) as f:
            node = int(f.read().strip())
        if node < 0:
            return None
        with open("/sys/devices/system/node/node%d/cpulist" % node) as f:
            cpus = parse_cpulist(f.read())
        allowed = os.sched_getaffinity(0)
        cpus = sorted(set(cpus) & allowed)
        return cpus or None
    except Exception:  # noqa: BLE001
        return None


def parse_cpulist(text):
    """'0-3,8,10-11' -> [0, 1, 2, 3, 8, 10, 11]"""
    out = []
    for part in text.strip().split(","):
        if not part:
            continue
        if "-" in part:
            lo, hi = part.split("-")
            out.extend(range(int(lo), int(hi) + 1))
        else:
            out.append(int(part))
    return out
