#!/usr/bin/env python3
"""bench.py -- frames/s of the per-frame super-resolution hot path on MI355X.

    python bench.py --gpus 1 --steps K --warmup W
    python -m torch.distributed.run --nnodes=1 --nproc-per-node N --master-addr 127.0.0.1 \
        --master-port P bench.py --gpus N --steps K --warmup W

A step = one pass of the hot path over one synthetic u8 BGR frame that is already resident in
HBM (upscale_image arithmetic: head/trunk/tail kernels over all reference tiles, u8 out in HBM).
N > 1: one process per GPU, frames sharded across ranks as independent units, no data-path
collective (RCCL is used only for the timing barrier / max-reduce); scaling = weak.

`value` is the device-resident rate (route "K": frames and results in HBM when the timed region
starts).  The PCIe-inclusive rate of the pipelined host route (route "E": page-locked host frames in,
page-locked host results out, H2D / kernels / D2H of consecutive frames overlapped) is measured on
EVERY rank in a second barrier-bracketed region and reported as config.host_route_fps_pcie_inclusive
(sum of frames over ranks / max wall over ranks): it is what shows the host-side limits of multi-GPU
scaling (feeder cores, pinned bandwidth, PCIe, NUMA), never `value`.  Every rank pins itself to the
CPUs of its GPU's NUMA node before it allocates its page-locked buffers.

One JSON line on rank 0 with the driver's keys plus
  roofline     : dominant kernel (trunk conv3x3 64->64) vs the dense fp16 MFMA peak, from HIP
                 events recorded on the engine's own stream inside the timed region
  cpu_baseline : the CPU oracle ("port": ncnn is not installable here) on a bounded sample
"""
import argparse
import json
import os
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.abspath(__file__))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)

MFMA_F16_DENSE_PEAK_TFLOPS = 2500.0   # /opt/skills/guides/MI355X_MICROARCH.md: ~2.5 PF dense fp16/bf16
# What the matrix pipes alone sustain at the 1400 W package cap with non-zero operands (nothing but
# register-operand v_mfma_f32_16x16x32_f16 on all 1024 SIMDs; tools/mfma_power_bench.hip, measured in
# profiles/r01_d_mfma_power_bench.txt).  The trunk kernel runs with the package pinned at the cap, so
# this -- not the 2.4 GHz figure above -- is the roof it can actually approach.  Informational only:
# `peak` / `frac` stay on the nominal figure.
MFMA_F16_SUSTAINED_AT_POWER_CAP_TFLOPS = 1880.0

WORKLOADS = {
    # name: (model key, model file stem, height, width)  -- BASELINE.json configs[1..4]
    "2x_compact_1080p": ("2x", "2x_Compact_Pretrain", 1080, 1920),
    "4x_compact_1080p": ("4x", "4x_Compact_Pretrain", 1080, 1920),
    "1x_hurrdeblur_1080p": ("1x", "1x_HurrDeblur_SubCompact_nf24-nc8_244k_net_g", 1080, 1920),
    "2x_compact_2160p": ("2x", "2x_Compact_Pretrain", 2160, 3840),
    # BASELINE.json configs[2]: 1x HurrDeblur -> u8 -> 2x Compact, both on the device
    "chain_1x_2x_1080p": ("2x", "2x_Compact_Pretrain", 1080, 1920),
    # BASELINE.json configs[3] as named: 4x_Valar_v1 (`-m r`) through the generic graph executor.  Its .bin is a missing
    # blob upstream (/root/reference/.MISSING_LARGE_BLOBS:1): random-init weights of that architecture unless the file is there
    "4x_valar_1080p": ("valar", "4x_Valar_v1", 1080, 1920),
}
VALAR_FLOP_PER_INPUT_PIXEL = 36_136_320      # SURVEY.md 8d (un-tiled frame)


def conv_flops_per_px(nf, nconv, scale):
    """2 * sum_conv(9*Cin*Cout) per input pixel."""
    return 2 * 9 * (3 * nf + (nconv - 2) * nf * nf + nf * 3 * scale * scale)


def shard_frames(n_frames, rank, world):
    """Frames are independent units: rank r takes frames r, r+world, ... (no exchange step)."""
    return list(range(rank, n_frames, world))


def timed_region(run_steps, sync, barrier, max_over_ranks, own=None):
    """barrier + sync, run, sync + barrier; returns the MAX wall seconds over ranks.  `own` (a list) also receives this rank's
    seconds from the common start to its OWN last result (before the closing barrier): the per-rank records' rate."""
    sync()
    barrier()
    t0 = time.perf_counter()
    run_steps()
    sync()
    mine = time.perf_counter() - t0
    barrier()
    if own is not None:
        own.append(mine)
    return max_over_ranks(time.perf_counter() - t0)


class FileCounter:
    """One counter for all ranks of a node (--dynamic: the frame queue of SURVEY.md 8e -- every rank takes the next frame index
    when it is ready for one, as the reference's pool hands out tasks, upscale/upscale_processing.py:565-601 -- instead of
    shard_frames' fixed r, r + world, ...): eight bytes in a file under /dev/shm, fetch-and-add under flock.  It works the same for
    self-launched ranks and for torch.distributed.run's, needs no collective, and costs ~2 us per take."""

    def __init__(self, path, create=False):
        import fcntl
        self._fcntl = fcntl
        self.path = path
        self._fd = os.open(path, os.O_RDWR | (os.O_CREAT if create else 0), 0o600)
        if create:
            self.reset()

    def reset(self, value=0):
        self._fcntl.flock(self._fd, self._fcntl.LOCK_EX)
        try:
            os.pwrite(self._fd, int(value).to_bytes(8, "little"), 0)
        finally:
            self._fcntl.flock(self._fd, self._fcntl.LOCK_UN)

    def take(self, n=1):
        """-> the counter's value before `n` was added"""
        self._fcntl.flock(self._fd, self._fcntl.LOCK_EX)
        try:
            v = int.from_bytes(os.pread(self._fd, 8, 0) or b"\0", "little")
            os.pwrite(self._fd, (v + n).to_bytes(8, "little"), 0)
        finally:
            self._fcntl.flock(self._fd, self._fcntl.LOCK_UN)
        return v

    def close(self):
        os.close(self._fd)


def job_scratch_dir():
    """where the ranks of THIS job meet on the node's file system (per-rank records, --dynamic's counter): launch_ranks makes it
    for self-launched ranks; under torch.distributed.run it is named after the rendezvous port"""
    d = os.environ.get("UVA_BENCH_JOB_DIR")
    if d:
        os.makedirs(d, exist_ok=True)
        return d
    import tempfile
    name = "uva_bench_%s_%s" % (os.environ.get("MASTER_PORT", "solo"), os.environ.get("TORCHELASTIC_RUN_ID", os.getppid()))
    for base in ("/dev/shm", tempfile.gettempdir(), ROOT):          # (the first one this user may write to)
        try:
            d = os.path.join(base, name)
            os.makedirs(d, exist_ok=True)
            if os.access(d, os.W_OK):
                return d
        except OSError:
            continue
    raise SystemExit("bench.py: no writable directory for the ranks' records (/dev/shm, %s, %s)" % (tempfile.gettempdir(), ROOT))


def gather_records(comm, record):
    """every rank's dict -> the list of all of them, in rank order, on every rank (through the job's scratch directory and two
    fences: no collective, any launcher)"""
    if comm.world == 1:
        return [record]                 # (one rank: nothing to exchange, no file system involved)
    d = job_scratch_dir()
    tmp = os.path.join(d, "rank%d.json.tmp" % comm.rank)
    with open(tmp, "w") as f:
        json.dump(record, f)
    os.replace(tmp, os.path.join(d, "rank%d.json" % comm.rank))
    comm.barrier()
    out = []
    for r in range(comm.world):
        with open(os.path.join(d, "rank%d.json" % r)) as f:
            out.append(json.load(f))
    comm.barrier()                      # nobody rewrites its file before everybody has read
    return out


def per_rank_summary(records):
    """min / max / mean of every numeric field over the ranks (what to look at first when 8 GPUs give less than 8 x)"""
    out = {}
    for key in sorted({k for r in records for k, v in r.items() if isinstance(v, (int, float)) and not isinstance(v, bool)}):
        if key in ("rank", "device", "numa_node"):
            continue
        vals = [r[key] for r in records if isinstance(r.get(key), (int, float))]
        if vals:
            out[key] = {"min": round(min(vals), 3), "max": round(max(vals), 3), "mean": round(sum(vals) / len(vals), 3)}
    return out


def gpu_numa_node(gpu):
    """NUMA node of the GPU's PCIe function (sysfs), or None"""
    try:
        import ctypes
        from upscale_video_amd import _lib
        buf = ctypes.create_string_buffer(64)
        if _lib.load().uva_get_gpu_pci_bus_id(int(gpu), buf, 64) != 0:
            return None
        with open("/sys/bus/pci/devices/%s/numa_node" % buf.value.decode().lower()) as f:
            return int(f.read().strip())
    except Exception:  # noqa: BLE001
        return None


def pcie_rates(torch, sync, barrier, mbytes=128, reps=3):
    """(h2d GB/s, d2h GB/s) of this rank's page-locked copies while EVERY rank copies at once (fenced): what the host route has
    per GPU when all of them pull on the node's memory and PCIe root complexes together"""
    n = mbytes << 20
    host = torch.empty(n, dtype=torch.uint8, pin_memory=True)
    dev = torch.empty(n, dtype=torch.uint8, device="cuda")
    out = []
    for src, dst in ((host, dev), (dev, host)):
        dst.copy_(src, non_blocking=True)
        sync()
        barrier()
        t0 = time.perf_counter()
        for _ in range(reps):
            dst.copy_(src, non_blocking=True)
        torch.cuda.synchronize()
        out.append(n * reps / (time.perf_counter() - t0) / 1e9)
        barrier()
    return out[0], out[1]


def whole_job_rate(units_per_rank, world, max_elapsed):
    """Aggregate over the job: every rank processed units_per_rank units, the slowest rank's wall time counts."""
    return units_per_rank * world / max_elapsed


def cpu_baseline(model_key, h, w, tile, min_seconds=10.0):
    """CPU oracle on a bounded sample of the same workload: crops of 1/16, 1/4, 1/1 of the frame
    are run in turn until one pass takes >= min_seconds (about 10-30 s of CPU work on the box's
    cores); the last one is reported, scaled to whole frames."""
    from oracle import uvoracle
    m = uvoracle.load_model(model_key)
    threads = uvoracle.max_threads()
    dt = frac = sh = sw = 0
    for div in (4, 2, 1):
        sh, sw = max(8, h // div), max(8, w // div)
        img = uvoracle.synthetic_frame(sh, sw)
        t0 = time.perf_counter()
        if tile > 0:
            m.upscale_image(img, tile_size=tile, border=10, threads=threads)
        else:
            m.apply_model(img, threads=threads)
        dt = time.perf_counter() - t0
        frac = (sh * sw) / float(h * w)
        if dt >= min_seconds or dt * 4 > 40:
            break
    # the same oracle on ONE core, on a crop that takes a few seconds (SURVEY.md 8d asks for both figures)
    sh1, sw1 = max(8, h // 4), max(8, w // 4)
    img1 = uvoracle.synthetic_frame(sh1, sw1)
    t0 = time.perf_counter()
    if tile > 0:
        m.upscale_image(img1, tile_size=tile, border=10, threads=1)
    else:
        m.apply_model(img1, threads=1)
    dt1 = time.perf_counter() - t0
    one_core = (sh1 * sw1) / float(h * w) / dt1
    return {
        "value": round(frac / dt, 5), "unit": "frames/s", "cores": threads, "kind": "port",
        "value_one_core": round(one_core, 6), "sample_one_core": f"one {sw1}x{sh1} crop, 1 thread, {dt1:.1f} s",
        "sample": f"one {sw}x{sh} crop ({frac:.4f} of a frame) through oracle/oracle.c (fp32, OpenMP), "
                  f"{dt:.1f} s, scaled to whole frames; ncnn itself is not installable here",
    }


def cpu_baseline_generic(param_path, bin_path, h, w, min_seconds=8.0):
    """The generic graphs' CPU restatement (oracle/generic_oracle.py: numpy fp32, whatever BLAS threads numpy has) on square
    crops of growing size until one pass takes >= min_seconds; scaled to whole frames by pixels (the arithmetic per pixel
    does not depend on the frame size)."""
    import numpy as np
    from oracle import generic_oracle, uvoracle
    m = generic_oracle.Model(param_path, bin_path)
    dt = side = 0
    for side in (24, 48, 96, 192):
        img = uvoracle.synthetic_frame(side, side)
        x = img.transpose(2, 0, 1).astype(np.float32) * np.float32(1 / 255.0)
        t0 = time.perf_counter()
        m.forward(x)
        dt = time.perf_counter() - t0
        if dt >= min_seconds or dt * 4 > 40:
            break
    frac = side * side / float(h * w)
    return {
        "value": round(frac / dt, 7), "unit": "frames/s", "cores": uvoracle.max_threads(), "kind": "port",
        "sample": f"one {side}x{side} crop ({frac:.6f} of a frame) through oracle/generic_oracle.py (numpy fp32, its BLAS threads), "
                  f"{dt:.1f} s, scaled to whole frames by pixels",
    }


def pmc_traffic_valar(launches_per_frame):
    """HBM bytes per rdb4_kernel launch from the committed PMC summary of the Valar frame (tools/pmc_valar.sh: FETCH_SIZE
    doubled per the gfx950 correction + WRITE_SIZE, per frame, summed over the kernel's launches) -> (bytes, file) or (None, None)."""
    import glob
    import re
    for path in reversed(sorted(glob.glob(os.path.join(ROOT, "profiles", "r*_valar_pmc.txt")))):
        try:
            fetch = write = None
            mode = None
            for line in open(path):
                if line.startswith("FETCH_SIZE"):
                    mode = "f"
                elif line.startswith("WRITE_SIZE"):
                    mode = "w"
                elif line.startswith(("HBM", "SQ")):
                    mode = None
                m = re.match(r"\s+uva::rdb4_kernel\(uva::RdbArgs\)\s+([0-9.]+) GB", line)
                if m and mode == "f" and fetch is None:
                    fetch = float(m.group(1))
                elif m and mode == "w" and write is None:
                    write = float(m.group(1))
            if fetch is not None and write is not None and launches_per_frame > 0:
                return int((2 * fetch + write) * 1e9 / launches_per_frame), "profiles/" + os.path.basename(path)
        except Exception:  # noqa: BLE001
            pass
    return None, None


def pmc_traffic(args, nf, kernel):
    """HBM bytes per launch of the dominant kernel from rocprofv3 PMC passes (FETCH_SIZE, WRITE_SIZE
    collected in separate runs of this command on the same build, FETCH_SIZE doubled per
    MI355X_MICROARCH.md's gfx950 correction).  bench.py cannot run the profiler on itself, so the
    committed summary of the matching workload AND kernel is replayed (-> (bytes, file name)), or
    (None, None) when there is none."""
    import glob
    if kernel == "sub10_kernel" and args.workload == "1x_hurrdeblur_1080p" and args.tile == 0:
        for path in reversed(sorted(glob.glob(os.path.join(ROOT, "profiles", "r*_sub10_pmc.json")))):
            try:
                return int(json.load(open(path))["hbm_bytes_per_launch"]), "profiles/" + os.path.basename(path)
            except Exception:  # noqa: BLE001
                pass
        return None, None
    paths = sorted(glob.glob(os.path.join(ROOT, "profiles", "r*_trunk_pmc.json")))   # latest round/letter last
    if args.workload != "2x_compact_1080p" or args.tile != 960 or nf != 64:
        return None, None
    for path in reversed(paths):
        try:
            d = json.load(open(path))
            if kernel + "<" in d.get("kernel", "trunk_kernel<"):
                return int(d["hbm_bytes_per_launch"]), "profiles/" + os.path.basename(path)
        except Exception:  # noqa: BLE001
            pass
    return None, None


def pin_to_gpu_numa_node(gpu):
    """CPUs of the NUMA node the GPU hangs off (sysfs via the engine's PCI bus id); None if unknown."""
    from upscale_video_amd import frame_pool
    cpus = frame_pool.gpu_cpu_affinity(gpu)
    if cpus:
        try:
            os.sched_setaffinity(0, cpus)
        except OSError:
            return None
    return cpus


def parity_probe(net, model_key, tile):
    """GPU vs CPU oracle on a small frame, reported next to the throughput."""
    from oracle import uvoracle
    img = uvoracle.synthetic_frame(96, 128)
    m = uvoracle.load_model(model_key)
    want = m.upscale_image(img, tile_size=64, border=10) if tile > 0 else m.apply_model(img)
    got = net.process_u8(img, tile_size=64 if tile > 0 else 0, border=10)
    d = got.astype(np.float64) - want.astype(np.float64)
    mse = float((d * d).mean())
    return {"psnr_db": round(99.0 if mse == 0 else 10 * np.log10(255.0 ** 2 / mse), 2),
            "max_abs_lsb": int(np.abs(d).max()), "vs": "CPU oracle fp32, 128x96 frame"}


def parity_windows(model_key, img, got, scale, rad, tile, border=10, win=48):
    """The TIMED workload's own frame against the fp32 oracle, in windows: every sample of nine `win`-pixel windows of the
    full-size result -- the four corners, the centre, and both sides of the reference tiling's seams -- is compared with the oracle
    run on the window's receptive field (`rad` pixels = one per 3x3 layer), cut to the window's TILE where the reference tiles
    (process_tile: each tile is its own zero-padded image, upscale/upscale_processing.py:395-477).  VERDICT r4 item 4: the
    128x96 probe says little about 1080p; a whole frame through the fp32 oracle takes 10 s of 16 cores, nine windows a second."""
    from oracle import uvoracle
    m = uvoracle.load_model(model_key)
    h, w = img.shape[:2]
    T = tile if tile > 0 else max(h, w)
    spots = [(0, 0), (0, w - win), (h - win, 0), (h - win, w - win), ((h - win) // 2, (w - win) // 2)]
    if tile > 0 and w > T:
        spots += [(h // 3, T - win), (h // 3, T)]              # either side of the x seam
    if tile > 0 and h > T:
        spots += [(T - win, w // 3), (T, w // 3)]              # ... and of the y seam
    worst, nz, se, n = 0, 0, 0.0, 0
    for (y0, x0) in spots:
        y0, x0 = max(0, min(y0, h - win)), max(0, min(x0, w - win))
        ty, tx = y0 // T, x0 // T
        y1, x1 = min(y0 + win, min((ty + 1) * T, h)), min(x0 + win, min((tx + 1) * T, w))      # (stay inside the tile's core)
        ry0, rx0, ry1, rx1 = ty * T, tx * T, min((ty + 1) * T, h), min((tx + 1) * T, w)
        if tile > 0:                                              # the tile with its borders (reference :409-427)
            ry0 -= border if ry0 >= border else 0
            rx0 -= border if rx0 >= border else 0
            ry1 += border if ry1 <= h - border else 0
            rx1 += border if rx1 <= w - border else 0
        cy0, cx0, cy1, cx1 = max(ry0, y0 - rad), max(rx0, x0 - rad), min(ry1, y1 + rad), min(rx1, x1 + rad)
        want = m.apply_model(np.ascontiguousarray(img[cy0:cy1, cx0:cx1]))[(y0 - cy0) * scale:(y1 - cy0) * scale, (x0 - cx0) * scale:(x1 - cx0) * scale]
        d = np.abs(got[y0 * scale:y1 * scale, x0 * scale:x1 * scale].astype(np.int16) - want.astype(np.int16))
        worst, nz, se, n = max(worst, int(d.max())), nz + int((d > 0).sum()), se + float((d.astype(np.float64) ** 2).sum()), n + d.size
    return {"psnr_db": round(99.0 if se == 0 else 10 * np.log10(255.0 ** 2 / (se / n)), 2), "max_abs_lsb": worst,
            "differ_share": round(nz / n, 5), "windows": len(spots), "samples": n,
            "vs": f"CPU oracle fp32 on the receptive fields of {len(spots)} {win}-px windows of the timed {w}x{h} frame "
                  f"(corners, centre, both sides of the tile seams)"}


def chain_parity_probe(pre, net):
    """1x -> u8 -> 2x on the GPU vs the same chain through the CPU oracle (config 3)."""
    from oracle import uvoracle
    img = uvoracle.synthetic_frame(96, 128)
    want = uvoracle.load_model("2x").upscale_image(uvoracle.load_model("1x").apply_model(img), tile_size=64, border=10)
    got = net.process_u8(pre.process_u8(img, tile_size=0), tile_size=64, border=10)
    d = got.astype(np.float64) - want.astype(np.float64)
    mse = float((d * d).mean())
    return {"psnr_db": round(99.0 if mse == 0 else 10 * np.log10(255.0 ** 2 / mse), 2),
            "max_abs_lsb": int(np.abs(d).max()), "vs": "CPU oracle fp32 chain 1x->u8->2x, 128x96 frame"}


class SoloComm:
    """one process, one GPU"""
    world, rank = 1, 0

    def barrier(self):
        pass

    def max_over_ranks(self, x):
        return x

    def close(self):
        pass


class RankLost(RuntimeError):
    """a rank of a self-launched job died or never arrived: the fence was aborted instead of waiting for ever"""


class ForkComm:
    """N ranks started by this script itself (python bench.py --gpus N): a multiprocessing.Barrier and a shared array --
    the frame queue has no data-path collective, and its timing fence does not need RCCL either.  A fence never waits for
    ever: the parent aborts the barrier as soon as a rank has exited (launch_ranks), and every wait has a timeout of its
    own as the backstop (UVA_BENCH_BARRIER_TIMEOUT seconds, default 900) -- either way the rank raises RankLost."""

    def __init__(self, rank, world, barrier, slots, timeout=None):
        self.rank, self.world, self._b, self._s = rank, world, barrier, slots
        self._timeout = float(os.environ.get("UVA_BENCH_BARRIER_TIMEOUT", "900")) if timeout is None else timeout

    def barrier(self):
        import threading
        try:
            self._b.wait(self._timeout)
        except threading.BrokenBarrierError:
            raise RankLost("rank %d: another rank left the job (or a fence timed out after %.0f s)" % (self.rank, self._timeout)) from None

    def max_over_ranks(self, x):
        self._s[self.rank] = x
        self.barrier()
        m = max(self._s[:self.world])
        self.barrier()          # nobody overwrites its slot before everybody has read
        return m

    def close(self):
        pass


class TorchComm:
    """launched by torch.distributed.run (the driver's N > 1 form): RCCL process group, used for the fence only"""

    def __init__(self, local_rank):
        import torch
        import torch.distributed as dist
        self._torch, self._dist = torch, dist
        dist.init_process_group("nccl", device_id=torch.device("cuda", local_rank))
        self.world, self.rank = dist.get_world_size(), dist.get_rank()

    def barrier(self):
        self._dist.barrier()

    def max_over_ranks(self, x):
        t = self._torch.tensor([x], dtype=self._torch.float64, device="cuda")
        self._dist.all_reduce(t, op=self._dist.ReduceOp.MAX)
        return float(t.item())

    def close(self):
        self._dist.barrier()
        self._dist.destroy_process_group()


def _fork_rank(rank, world, device, barrier, slots, argv):
    """entry point of a self-launched rank (spawn context: a fresh interpreter)"""
    sys.argv = argv
    run(parse_args(argv[1:]), ForkComm(rank, world, barrier, slots), device)


def launch_ranks(world, devices, argv, target=None, poll_s=0.2, grace_s=5.0):
    """python bench.py --gpus N: start the N ranks here (spawn context, one process per GPU, no RCCL anywhere) and WATCH them:
    the first rank that exits with a non-zero code -- an out-of-memory kill while page-locking its rings, a missing device 7,
    a HIP error, an assert -- aborts the fence (the others leave their wait with RankLost instead of sitting in it until the
    driver's timeout), the rest get `grace_s` seconds to go and are then terminated, and the job's exit code is that rank's.
    Returns 0 or the first failing exit code (negative = killed by that signal); prints what happened to stderr."""
    import multiprocessing as mp
    import tempfile
    ctx = mp.get_context("spawn")
    barrier, slots = ctx.Barrier(world), ctx.Array("d", world)
    target = target or _fork_rank
    if not os.environ.get("UVA_BENCH_JOB_DIR"):     # (the ranks inherit the environment: per-rank records, --dynamic's counter)
        os.environ["UVA_BENCH_JOB_DIR"] = tempfile.mkdtemp(prefix="uva_bench_", dir="/dev/shm" if os.path.isdir("/dev/shm") else None)
    procs = [ctx.Process(target=target, args=(r, world, devices[r], barrier, slots, list(argv))) for r in range(world)]
    for p in procs:
        p.start()
    failed = None
    while any(p.exitcode is None for p in procs):
        for r, p in enumerate(procs):
            p.join(poll_s / world)
            if p.exitcode not in (None, 0) and failed is None:
                failed = (r, p.exitcode)
        if failed is not None:
            break
    if failed is None:
        bad = [(r, p.exitcode) for r, p in enumerate(procs) if p.exitcode != 0]
        if not bad:
            return 0
        failed = bad[0]
    barrier.abort()                     # every rank waiting at (or arriving at) a fence raises RankLost now
    deadline = time.monotonic() + grace_s
    for p in procs:
        p.join(max(0.0, deadline - time.monotonic()))
    for p in procs:
        if p.exitcode is None:
            p.terminate()
    for p in procs:
        p.join(2.0)
        if p.exitcode is None:
            p.kill()
    r, code = failed
    what = "was killed by signal %d" % -code if code < 0 else "exited with code %d" % code
    print("bench.py: rank %d (device %s) %s; the fence was aborted and the other %d rank(s) stopped -- no result line"
          % (r, devices[r], what, world - 1), file=sys.stderr, flush=True)
    return code


def pinned_budget_check(world, ranks_here, h, w, scale, depth=3):
    """Page-locked host memory the job's ranks on THIS node are about to take (the host route's rings: `depth` frames in and
    `depth` results out per rank) against what the node can lock: MemAvailable and, in a container, the cgroup's memory.max.
    Eight ranks at 3840x2160 -> 2x want 8 x 3 x (24.9 + 99.5) MB = 3.0 GB; a rank that dies in hipHostMalloc half way through
    takes the job down (launch_ranks sees it), this says so before anything is allocated.  Returns (needed, available) bytes."""
    per_rank = depth * (h * w * 3 + (h * scale) * (w * scale) * 3)
    needed = per_rank * ranks_here
    avail = None
    try:
        for line in open("/proc/meminfo"):
            if line.startswith("MemAvailable:"):
                avail = int(line.split()[1]) * 1024
    except OSError:
        pass
    try:
        lim = open("/sys/fs/cgroup/memory.max").read().strip()
        if lim != "max":
            used = int(open("/sys/fs/cgroup/memory.current").read())
            avail = min(avail, int(lim) - used) if avail is not None else int(lim) - used
    except (OSError, ValueError):
        pass
    if avail is not None and needed > 0.8 * avail:
        raise SystemExit("bench.py: the %d rank(s) of this node need %.2f GB of page-locked host memory (%d frames in flight per rank) "
                         "and the node can give %.2f GB: fewer ranks, a smaller frame, or more memory"
                         % (ranks_here, needed / 1e9, depth, avail / 1e9))
    return needed, avail


def parse_args(argv=None):
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=200)
    ap.add_argument("--warmup", type=int, default=20)
    ap.add_argument("--workload", default="2x_compact_1080p", choices=sorted(WORKLOADS))
    ap.add_argument("--tile", type=int, default=None,
                    help="reference tile size; 0 = whole frame.  Default: what the reference does for the workload's model -- "
                         "960 (upscale_image, :499-516) for the 2x / 4x nets, 0 (apply_model, whole frame, :263-288) for the 1x net")
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--no-parity", action="store_true",
                    help="skip the 128x96 parity probe (profiling runs: every launch in the trace is then a full-size one)")
    ap.add_argument("--devices", default=None,
                    help="comma-separated device ordinal per rank (default 0,1,..,N-1); '0,0' puts two ranks on one GPU, the "
                         "reference's own way of loading a GPU with several workers (README.md:45-61)")
    ap.add_argument("--dynamic", action="store_true",
                    help="N > 1: the ranks pull frame indices from ONE shared counter (the reference's pool queue, "
                         "upscale_processing.py:565-601) instead of owning --steps frames each; the job is still steps x N frames")
    ap.add_argument("--batch", type=int, default=None,
                    help="frames per uva_net_process_u8_device_batch call for the workloads with the 1x net (default 4; 1 = one "
                         "call per frame).  Ignored for the other workloads: their nets run frame by frame either way")
    ap.add_argument("--repeats", type=int, default=3, help="timed regions of --steps steps each; the median is reported (SURVEY.md 8d)")
    return ap.parse_args(argv)


def library_record():
    """Which libuva.so this process measures: path, short hash, and whether build_lib() compiled it here or found it fresh
    (the GPU box gets the library with the snapshot; VERDICT r3 weak 11 asked for the record)."""
    import hashlib
    from upscale_video_amd import build
    path = os.environ.get("UVA_LIB_PATH") or build.LIB
    try:
        with open(path, "rb") as f:
            digest = hashlib.sha256(f.read()).hexdigest()[:16]
    except OSError:
        digest = None
    return {"file": os.path.relpath(path, ROOT), "sha256_16": digest, "compiled_by_this_run": bool(_LIB_BUILT_HERE)}


_LIB_BUILT_HERE = False


def main():
    global _LIB_BUILT_HERE
    args = parse_args()
    from upscale_video_amd import build
    _LIB_BUILT_HERE = not os.environ.get("UVA_LIB_PATH") and build.needs_build()
    build.build_lib()                      # once, before any rank exists
    if "RANK" in os.environ:               # torch.distributed.run started us (the driver's form for N > 1, also legal with 1 rank)
        local_rank = int(os.environ.get("LOCAL_RANK", "0"))
        devices = [int(d) for d in args.devices.split(",")] if args.devices else None
        device = devices[local_rank] if devices else local_rank
        import torch
        torch.cuda.set_device(device)
        comm = TorchComm(device)
        if comm.world != args.gpus and comm.rank == 0:
            print("bench.py: --gpus %d but WORLD_SIZE is %d: using the world size" % (args.gpus, comm.world), file=sys.stderr)
        run(args, comm, device)
        return
    devices = [int(d) for d in args.devices.split(",")] if args.devices else list(range(args.gpus))
    if len(devices) != args.gpus:
        sys.exit("--devices names %d devices for --gpus %d" % (len(devices), args.gpus))
    if args.gpus == 1:
        run(args, SoloComm(), devices[0])
        return
    rc = launch_ranks(args.gpus, devices, list(sys.argv))
    if rc:
        sys.exit(rc if 0 < rc < 256 else 1)


def run(args, comm, device):
    import torch
    from upscale_video_amd import ncnn
    from upscale_video_amd.synth import synthetic_frame, synthetic_weights

    world, rank, local_rank = comm.world, comm.rank, device
    assert torch.cuda.is_available(), "bench.py needs an MI355X (no CPU path)"
    torch.cuda.set_device(local_rank)
    barrier, max_over_ranks = comm.barrier, comm.max_over_ranks

    numa_cpus = pin_to_gpu_numa_node(local_rank)     # before any page-locked allocation of this rank
    key, stem, h, w = WORKLOADS[args.workload]
    if args.tile is None:
        args.tile = 0 if key == "1x" else 960
    net = ncnn.Net()
    net.set_vulkan_device(local_rank)
    base = os.path.join(ROOT, "models", stem)
    generic = key == "valar"
    weights, weights_note = base + ".bin", "the repository's .bin"
    if generic and not os.path.exists(weights):
        import tempfile
        tmpdir = tempfile.TemporaryDirectory()
        weights = os.path.join(tmpdir.name, stem + ".bin")
        synthetic_weights(base + ".param", weights, seed=1, gain=0.5)
        weights_note = "random-init (the .bin is a missing blob upstream): throughput only"
    assert net.load_param(base + ".param") == 0 and net.load_model(weights) == 0, getattr(net, "last_error", "")
    s = net.scale

    # synthetic frames, resident in HBM before the timed region
    n_src = 4
    frames = [torch.from_numpy(synthetic_frame(h, w, seed=20260928 + 17 * rank + i)).cuda() for i in range(n_src)]
    out = torch.empty((h * s, w * s, 3), dtype=torch.uint8, device="cuda")
    torch.cuda.synchronize()
    # frames per call for the workloads with the 1x net (uva_net_process_u8_device_batch: up to eight frames per sub10_kernel launch)
    batch = max(1, min(8, args.batch if args.batch else 4)) if (key == "1x" or args.workload.startswith("chain_")) else 1
    outs = [out] + [torch.empty_like(out) for _ in range(batch - 1)] if key == "1x" else [out]

    pre = None
    if args.workload.startswith("chain_"):
        pre = ncnn.Net()      # the '-m a' pass: whole frame, no tiling (apply_model)
        pre.set_vulkan_device(local_rank)
        pbase = os.path.join(ROOT, "models", "1x_HurrDeblur_SubCompact_nf24-nc8_244k_net_g")
        assert pre.load_param(pbase + ".param") == 0 and pre.load_model(pbase + ".bin") == 0
        mid = [torch.empty((h, w, 3), dtype=torch.uint8, device="cuda") for _ in range(max(2, batch))]
        torch.cuda.synchronize()

    def step(i):
        src = frames[i % n_src].data_ptr()
        if pre is not None:
            m = mid[i & 1]
            pre.wait_for(net)                     # mid[i&1] was last read two frames ago by `net`
            pre.process_u8_device(src, h, w, m.data_ptr(), tile_size=0)
            net.wait_for(pre)
            src = m.data_ptr()
        net.process_u8_device(src, h, w, out.data_ptr(), tile_size=args.tile, border=10)

    def step_group(i0, k):
        """frames i0 .. i0 + k - 1: one call per frame, or -- the workloads with the 1x net, --batch > 1 -- one batch call of the
        1x net for all k (the chain: k frames through the 1x net, then each through the 2x net)"""
        if batch == 1 or k == 1:
            for i in range(i0, i0 + k):
                step(i)
        elif pre is None:
            net.process_u8_device_batch([frames[(i0 + j) % n_src].data_ptr() for j in range(k)], h, w,
                                        [outs[j].data_ptr() for j in range(k)], tile_size=args.tile, border=10)
        else:
            pre.wait_for(net)                     # the mid buffers were last read by the previous group's 2x passes
            pre.process_u8_device_batch([frames[(i0 + j) % n_src].data_ptr() for j in range(k)], h, w,
                                        [mid[j].data_ptr() for j in range(k)], tile_size=0)
            net.wait_for(pre)
            for j in range(k):
                net.process_u8_device(mid[j].data_ptr(), h, w, out.data_ptr(), tile_size=args.tile, border=10)

    def run_static(n_frames):
        for i0 in range(0, n_frames, batch):
            step_group(i0, min(batch, n_frames - i0))

    counter = None
    if args.dynamic:
        cpath = os.path.join(job_scratch_dir(), "frame_counter")
        if rank == 0:
            FileCounter(cpath, create=True).close()
        barrier()
        counter = FileCounter(cpath)
    taken = [0]

    def run_dynamic(total):
        """every rank takes the next `chunk` frame indices off the one counter when its queue has drained (the launches are
        asynchronous: without the wait one rank would take the whole job into its stream in a millisecond)"""
        chunk = max(batch, 4)
        taken[0] = 0
        while True:
            i0 = counter.take(chunk)
            if i0 >= total:
                break
            k = min(chunk, total - i0)
            for j0 in range(0, k, batch):
                step_group(i0 + j0, min(batch, k - j0))
            taken[0] += k
            sync()

    def sync():
        if pre is not None:
            pre.synchronize()
        net.synchronize()
        torch.cuda.synchronize()

    t_first = time.perf_counter()
    step_group(0, batch)                          # the first call: workspace, schedules, row tables
    sync()
    first_frame_ms = (time.perf_counter() - t_first) * 1e3
    run_static(args.warmup)
    sync()
    # >= 3 timed regions of exactly --steps steps each (every one bracketed by barrier + sync, MAX over ranks); the MEDIAN region
    # counts -- for `value` and for the kernel event statistics alike: set_profiling(True) zeroes the counters, so every region
    # has its own, and the roofline's launch time is the reported region's (a process's first ~40 launches run while the clock
    # still ramps: with the driver's --steps 20 --warmup 5 they sit in the first region; profiles/r04_ab_results.txt block 21)
    regions, region_stats, own_s, own_frames = [], [], [], []
    for _ in range(max(1, args.repeats)):
        net.set_profiling(True)
        if counter is not None:
            if rank == 0:
                counter.reset()
            regions.append(timed_region(lambda: run_dynamic(args.steps * world), sync, barrier, max_over_ranks, own=own_s))
            own_frames.append(taken[0])
        else:
            regions.append(timed_region(lambda: run_static(args.steps), sync, barrier, max_over_ranks, own=own_s))
            own_frames.append(args.steps)
        region_stats.append([net.kernel_stats(k) for k in range(3)])
    net.set_profiling(False)
    median_idx = sorted(range(len(regions)), key=lambda i: regions[i])[len(regions) // 2]
    elapsed = regions[median_idx]
    (_, head_ms), (n_launch, trunk_ms), (n_tail, tail_ms) = region_stats[median_idx]   # kind 1: trunk launches (generic graphs:
                                                                                       # rdb4_kernel), kind 2: tail (conv5)

    fps = whole_job_rate(args.steps, world, elapsed)
    steps_timed = max(1, own_frames[median_idx])   # the kernel event statistics are the median region's, of THIS rank's frames

    # (E) pipelined host route, PCIe inclusive, on every rank at once: frames in page-locked host memory,
    # submit/collect with 3 frames in flight (SURVEY.md 8d "host-to-host with stream overlap")
    depth, n_host = 3, (max(6, min(args.steps, 12)) if generic else max(30, min(args.steps, 120)))
    pinned_need, pinned_avail = pinned_budget_check(world, world, h, w, s, depth)      # one node: every rank is here
    host_in = frames[0].cpu().numpy()
    pin_in = [ncnn.pinned_empty((h, w, 3)) for _ in range(depth)]
    pin_out = [ncnn.pinned_empty((h * s, w * s, 3)) for _ in range(depth)]
    for b in pin_in:
        b[...] = host_in

    def host_pipeline(n_frames):
        inflight = []
        for i in range(n_frames):
            if len(inflight) == depth:
                net.collect_u8(inflight.pop(0))
            inflight.append(net.submit_u8(pin_in[i % depth], out=pin_out[i % depth], tile_size=args.tile, border=10))
        while inflight:
            net.collect_u8(inflight.pop(0))

    host_fps = None
    own_host = []
    if pre is None:
        host_pipeline(6)
        host_elapsed = timed_region(lambda: host_pipeline(n_host), sync, barrier, max_over_ranks, own=own_host)
        host_fps = whole_job_rate(n_host, world, host_elapsed)

    # the 1x net one frame per launch beside the batch figure (the reference's unit is the frame: both are reported)
    single_fps = None
    if batch > 1 and pre is None:
        t_single = timed_region(lambda: [step(i) for i in range(args.steps)], sync, barrier, max_over_ranks)
        single_fps = whole_job_rate(args.steps, world, t_single)

    # per-rank records (when eight GPUs give less than 6 x, the one JSON line must say which rank, which
    # NUMA node, which PCIe direction): every rank's own rates, gathered through the job's scratch directory
    h2d, d2h = pcie_rates(torch, sync, barrier)
    record = {"rank": rank, "device": local_rank, "numa_node": gpu_numa_node(local_rank), "cpus_pinned": (len(numa_cpus) if numa_cpus else None),
              "frames_K": own_frames[median_idx], "fps_K": round(own_frames[median_idx] / own_s[median_idx], 2),
              "fps_E": (round(n_host / own_host[0], 2) if own_host else None),
              "h2d_GBps": round(h2d, 2), "d2h_GBps": round(d2h, 2), "first_frame_ms": round(first_frame_ms, 1)}
    per_rank = gather_records(comm, record)
    if counter is not None:
        counter.close()

    if rank == 0 and generic:
        # dominant kernel: rdb4_kernel, the first four convolutions (+ the 1x1) of a residual dense block for all planes of
        # the frame in one launch, timed by HIP events around every launch; 262 144 FLOP per input pixel
        frame_flops = VALAR_FLOP_PER_INPUT_PIXEL * h * w
        rdb_flops = 2 * (9 * 32 * (64 + 96 + 128 + 160) + 64 * 32) * h * w
        avg_ms = trunk_ms / max(1, n_launch)
        # per FRAME totals: a frame's planes may go through a dense block in several launches (plane batches), the frame's
        # 69 blocks' worth of FLOPs over the time of all their launches is what holds in every case
        VALAR_BLOCKS = 69
        achieved = rdb_flops * VALAR_BLOCKS * steps_timed / (trunk_ms * 1e-3) / 1e12 if trunk_ms > 0 else 0.0
        traffic, traffic_source = pmc_traffic_valar(n_launch / max(1, steps_timed)) if args.tile == 960 and (h, w) == (1080, 1920) else (None, None)
        result = {
            "metric": "frames/sec " + args.workload, "value": round(fps, 3), "unit": "frames/s", "n_gpus": world,
            "steps": args.steps, "warmup": args.warmup, "ms_per_step": round(elapsed / args.steps * 1e3, 4),
            "higher_is_better": True, "scaling": "weak", "vs_baseline": None, "dtype": "f16 operands / f32 accumulate",
            "data": "synthetic frames; weights: " + weights_note,
            "config": {
                "workload": f"{w}x{h} synthetic u8 BGR frames, {stem} (1 206 layers) through the generic graph executor, "
                            f"upscale_image arithmetic ({'reference 960-px tiles, 10-px border' if args.tile > 0 else 'whole frame'}), "
                            f"frames and results resident in HBM",
                "route": "K (device-resident frames and results; host_route_* fields are PCIe inclusive and never `value`)",
                "frames_per_rank": args.steps, "tile_size": args.tile, "parallelism": f"frame-sharded x{world}, no collective",
                "timed_regions_s": [round(x, 5) for x in regions], "reported": "median region (value and kernel event statistics)",
                "frame_tflop": round(frame_flops / 1e12, 4), "whole_path_tflops": round(frame_flops * fps / world / 1e12, 1),
                "kernel_ms_per_frame": {"rdb4_kernel": round(trunk_ms / steps_timed, 3), "conv5 (g_conv3_sww: 192 -> 64 as Winograd F(2,3))": round(tail_ms / steps_timed, 3)},
                "numa_cpus_rank0": (len(numa_cpus) if numa_cpus else None),
                "pinned_host_bytes_all_ranks": pinned_need, "host_memory_available_bytes": pinned_avail,
                "frame_queue": "dynamic (one shared counter)" if counter is not None else "static (frames r, r + N, ...)",
                "library": library_record(),
            },
            "per_rank": per_rank, "per_rank_summary": per_rank_summary(per_rank),
            "roofline": {"kernel": "rdb4_kernel (conv1..conv4 + the 1x1 of a residual dense block, every plane of the frame, one launch)",
                         "bound": "mfma", "achieved": round(achieved, 1), "peak": MFMA_F16_DENSE_PEAK_TFLOPS, "unit": "TFLOP/s",
                         "frac": round(achieved / MFMA_F16_DENSE_PEAK_TFLOPS, 4), "traffic": traffic, "traffic_source": traffic_source,
                         "flops_per_launch": rdb_flops, "avg_launch_ms": round(avg_ms, 4), "launches": n_launch,
                         "launches_per_frame": round(n_launch / max(1, steps_timed), 2)},
            "cpu_baseline": None,
        }
        if not args.no_cpu_baseline:
            result["cpu_baseline"] = cpu_baseline_generic(base + ".param", weights, h, w)
        if host_fps is not None:
            result["config"]["host_route_fps_pcie_inclusive"] = round(host_fps, 2)
            result["config"]["host_route_frames_per_rank"] = n_host
        print(json.dumps(result), flush=True)
    elif rank == 0:
        nf, nconv = net.num_features, net.num_convs
        layers_per_launch = (nconv - 2) * steps_timed / max(1, n_launch)      # 2 with the fused-pair kernels
        fused = layers_per_launch > 1.5
        # the fused pair runs as Winograd F(2,3) (trunkw_kernel) unless UVA_TRUNK_WINO=0 selects the direct trunk2_kernel
        kernel = (("trunkw_kernel" if not (os.environ.get("UVA_DEBUG_SWITCHES") == "1" and os.environ.get("UVA_TRUNK_WINO", "1") == "0") else "trunk2_kernel") if nf == 64 else "pair24_kernel") if fused else \
                 ("trunk_kernel" if nf == 64 else "conv3x3_kernel")
        trunk_flops_per_launch = layers_per_launch * 2 * 9 * nf * nf * h * w        # algorithmic: un-tiled frame
        whole_net = nf == 24 and layers_per_launch > nconv - 2.5    # sub10_kernel: all ten convolutions of the 1x net in one launch
        split5 = nf == 24 and 3.5 < layers_per_launch < 4.5         # sub5_kernel (UVA_SUB5=1): the same net as two launches of five layers
        frames_per_launch = steps_timed / max(1, n_launch) if whole_net else 1.0      # (uva_net_process_u8_device_batch: up to 8)
        if whole_net:
            trunk_flops_per_launch = conv_flops_per_px(nf, nconv, s) * h * w * frames_per_launch
        if split5:
            trunk_flops_per_launch = conv_flops_per_px(nf, nconv, s) * h * w / 2.0     # (the frame's convolutions over its two launches)
        avg_ms = trunk_ms / max(1, n_launch)
        achieved = trunk_flops_per_launch / (avg_ms * 1e-3) / 1e12 if avg_ms > 0 else 0.0
        frame_flops = conv_flops_per_px(nf, nconv, s) * h * w
        if pre is not None:
            frame_flops += conv_flops_per_px(pre.num_features, pre.num_convs, 1) * h * w
        traffic, traffic_source = pmc_traffic(args, nf, "sub10_kernel" if whole_net else kernel)
        if whole_net and traffic is not None:
            traffic = int(traffic * frames_per_launch)      # (the committed figure is one frame's launch: a batch moves as many frames' bytes)
        result = {
            "metric": "frames/sec 1080p->2x Compact (SRVGGNetCompact per-frame SR hot path)" if args.workload == "2x_compact_1080p"
                      else "frames/sec " + args.workload,
            "value": round(fps, 3), "unit": "frames/s", "n_gpus": world, "steps": args.steps, "warmup": args.warmup,
            "ms_per_step": round(elapsed / args.steps * 1e3, 4), "higher_is_better": True, "scaling": "weak",
            "vs_baseline": None, "dtype": "f16 operands / f32 accumulate", "data": "synthetic",
            "config": {
                "workload": f"{w}x{h} synthetic u8 BGR frames, {stem}, upscale_image arithmetic "
                            f"({'reference 960-px tiles, 10-px border' if args.tile > 0 else 'whole frame'}), "
                            f"frames and results resident in HBM",
                "route": "K (device-resident frames and results; host_route_* fields are PCIe inclusive and never `value`)",
                "frames_per_rank": args.steps, "tile_size": args.tile, "parallelism": f"frame-sharded x{world}, no collective",
                "timed_regions_s": [round(x, 5) for x in regions], "reported": "median region (value and kernel event statistics)",
                "launcher": {"SoloComm": "one process", "ForkComm": "python bench.py --gpus N: own processes, multiprocessing barrier, no RCCL",
                             "TorchComm": "torch.distributed.run, RCCL used for the timing fence only"}[type(comm).__name__],
                "device_rank0": local_rank,
                "frame_tflop": round(frame_flops / 1e12, 4),
                "whole_path_tflops": round(frame_flops * fps / world / 1e12, 1),
                "kernel_ms_per_frame": {"head": round(head_ms / steps_timed, 4), "trunk": round(trunk_ms / steps_timed, 4),
                                        "tail": round(tail_ms / steps_timed, 4)},
                "numa_cpus_rank0": (len(numa_cpus) if numa_cpus else None),
                "pinned_host_bytes_all_ranks": pinned_need, "host_memory_available_bytes": pinned_avail,
                "frame_queue": "dynamic (one shared counter)" if counter is not None else "static (frames r, r + N, ...)",
                "frames_per_call": batch,
                "library": library_record(),
            },
            "per_rank": per_rank, "per_rank_summary": per_rank_summary(per_rank),
            "roofline": {
                "kernel": (f"sub10_kernel (the whole 1x net: 3->24, 8 x 24->24, 24->3, + input; {frames_per_launch:.3g} frame(s) per launch)" if whole_net else
                           "sub5_kernel (the 1x net as two launches of five layers, two pipelines per workgroup; per launch: half the net's FLOPs)" if split5 else
                           (f"{kernel}<{nf}>" if nf == 64 else ("pair24_kernel" if fused else f"conv3x3_kernel<{nf},0,1>")) +
                           (f" ({int(round(layers_per_launch))} trunk layers {nf}->{nf} + PReLU per launch)")),
                "bound": "mfma", "achieved": round(achieved, 1), "peak": MFMA_F16_DENSE_PEAK_TFLOPS, "unit": "TFLOP/s",
                "frac": round(achieved / MFMA_F16_DENSE_PEAK_TFLOPS, 4), "traffic": traffic, "traffic_source": traffic_source,
                "flops_per_launch": trunk_flops_per_launch, "avg_launch_ms": round(avg_ms, 4), "launches": n_launch,
                "layers_per_launch": round(layers_per_launch, 2), "frames_per_launch": round(frames_per_launch, 3),
                # `achieved` / `frac` count the ALGORITHMIC FLOPs of the direct 3x3 convolution (SURVEY.md 8d's per-unit figure)
                # -- what the roofline contract asks for; trunkw_kernel ISSUES two thirds of them (1-D Winograd F(2,3): four
                # multiplications for two columns instead of six) and recomputes strip edges, so the matrix pipes' own utilisation is
                # lower: stated here so that nobody reads 0.5 as pipe occupancy
                "flops_kind": "algorithmic (direct convolution on the un-tiled frame)",
                "issued_over_algorithmic": (round(2.0 / 3.0, 4) if kernel == "trunkw_kernel" else 1.0) if nf == 64 and fused else None,
                "achieved_issued": (round(achieved * (2.0 / 3.0 if kernel == "trunkw_kernel" else 1.0), 1) if nf == 64 and fused else None),
                "frac_issued": (round(achieved * (2.0 / 3.0 if kernel == "trunkw_kernel" else 1.0) / MFMA_F16_DENSE_PEAK_TFLOPS, 4)
                                if nf == 64 and fused else None),
                "peak_sustained_at_power_cap": MFMA_F16_SUSTAINED_AT_POWER_CAP_TFLOPS,
                "frac_of_sustained": round(achieved / MFMA_F16_SUSTAINED_AT_POWER_CAP_TFLOPS, 4) if nf == 64 else None,
            },
        }
        if host_fps is not None:
            result["config"]["host_route_fps_pcie_inclusive"] = round(host_fps, 2)
            result["config"]["host_route_frames_per_rank"] = n_host
        if single_fps is not None:
            result["config"]["one_frame_per_call_fps"] = round(single_fps, 2)      # the same net through uva_net_process_u8_device
        if world == 1:
            # informational: pageable numpy in/out, one synchronous call per frame (what one reference worker does)
            net.process_u8(host_in, tile_size=args.tile, border=10)
            t0 = time.perf_counter()
            for _ in range(5):
                net.process_u8(host_in, tile_size=args.tile, border=10)
            result["config"]["host_route_sync_pageable_fps"] = round(5 / (time.perf_counter() - t0), 2)
            if pre is None:
                ref_out = net.process_u8(host_in, tile_size=args.tile, border=10)
                assert np.array_equal(pin_out[(n_host - 1) % depth], ref_out), "pipelined host route differs from the synchronous one"
            if not args.no_parity:
                result["parity"] = parity_probe(net, key, args.tile) if pre is None else chain_parity_probe(pre, net)
                if pre is None:
                    result["parity"]["full_size"] = parity_windows(key, host_in, ref_out, s, nconv, args.tile)
            if not args.no_cpu_baseline:
                result["cpu_baseline"] = cpu_baseline(key, h, w, args.tile)
        print(json.dumps(result), flush=True)
    comm.close()


if __name__ == "__main__":
    main()
