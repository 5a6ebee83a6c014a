"""rawvideo.py -- stream bgr24 frames through the MI355X engine.

The reference moves every frame through PNG files on both sides of the net (ffmpeg writes
`%d.extract.png`, upscale/upscale_processing.py:203-255; workers imread / imwrite, :263,288,487,519;
ffmpeg re-reads `%d.png`, :604-640) and the PNG codecs, not the GPU, bound its file-to-file rate.
This module is the same per-frame arithmetic fed by a raw pipe instead, ffmpeg itself untouched:

    ffmpeg -i in.mkv -f rawvideo -pix_fmt bgr24 - \\
      | python -m upscale_video_amd.rawvideo -W 1920 -H 1080 -s 2 \\
      | ffmpeg -f rawvideo -pix_fmt bgr24 -s 3840x2160 -r 24 -i - out.mkv

Frames go through Net.submit_u8 / collect_u8 (include/uva.h): page-locked rings on the host, H2D copy,
kernels and D2H copy of consecutive frames overlapped on three streams.  `-m a` runs the reference's
anime pass first (1x HurrDeblur, whole frame, re-quantised to u8 exactly like the PNG hop of
upscale_processing.py:909), then the 2x/4x net with the reference's 960-px tiles.

Flags follow upscale_video.py where they mean the same thing: -s/--scale 1|2|4, -m/--models a,n=K,r (in the reference's
order: denoise, anime pass, upscale; upscale/upscale_processing.py:880-920),
-g/--gpu a list of HIP ordinals, one worker per entry (duplicates allowed).  Pipes: one reader deals the frames
out round-robin, one writer puts the results out in frame order.  File to file: one contiguous segment of frames per
entry, each with its own reader, chain of nets and writer on its own file handles (stream_segments) -- no shared serial
copy, so the route scales with the GPUs.  Measured on the GPU box (profiles/r05_ab_results.txt blocks 8, 9): on a real file
system (the container's overlay) positional writes of several workers into ONE file scale -- 187 (one worker), 372 (two), 436
(four) frames/s at 1080p -> 4K = 0.93 of the /dev/null rate; on tmpfs (/dev/shm) one file takes ~200 frames/s = 5 GB/s whatever
writes into it and more writers make it worse -- its page allocation under the inode lock is the limit, and only `-o x,y,...`
(one file per worker: separate inodes, 457-477 frames/s) gets past it there.  Two things built to beat the inode lock and
measured no better, kept as options: MappedSegment (each worker copying through a shared mapping of its byte range,
UVA_RAW_MMAP=1: slower than write() on tmpfs, equal on overlay) and `--write-threads N` (a worker's frames cut into pieces
written with os.pwrite by a pool: +10 % for one worker on overlay, -40 % on tmpfs; default 1).
"""
import argparse
import ctypes
import mmap
import os
import sys
import threading
import queue

import numpy as np

from . import ncnn
from .upscale_processing import TILE_BORDER, TILE_SIZE

MODEL_FILES = {  # upscale/upscale_processing.py:880-916
    1: "1x_HurrDeblur_SubCompact_nf24-nc8_244k_net_g",
    2: "2x_Compact_Pretrain",
    4: "4x_Compact_Pretrain",
}
PIPE_DEPTH = 3   # include/uva.h: at most 3 frames in flight per net


def load_net(stem, gpu, model_path):
    net = ncnn.Net()
    net.opt.use_vulkan_compute = True
    net.set_vulkan_device(gpu)
    base = os.path.join(model_path, stem)
    if net.load_param(base + ".param") != 0 or net.load_model(base + ".bin") != 0:
        raise RuntimeError(getattr(net, "last_error", "cannot load " + base))
    return net


def read_exact(f, view):
    """Fill `view` from f; returns False on a clean EOF before the first byte, raises on a torn frame."""
    got = 0
    n = len(view)
    while got < n:
        k = f.readinto(view[got:])
        if not k:
            if got == 0:
                return False
            raise EOFError("input ended inside a frame (%d of %d bytes)" % (got, n))
        got += k
    return True


class Stage:
    """One net with its own ring of page-locked result buffers and up to PIPE_DEPTH frames in flight."""

    def __init__(self, net, h, w, tile_size, alloc):
        self.net, self.tile = net, tile_size
        s = net.scale
        # A result buffer stays in use while its frame is in flight here (<= depth frames) and then
        # while the consumer holds it -- the next stage reads it as pinned input until that stage's
        # collect (<= depth frames), or the writer thread (queue of 2 + 1 being written).  Frames stay
        # in order, so 2*depth + 2 buffers can never wrap onto a live one.
        self.outs = [alloc((h * s, w * s, 3)) for _ in range(2 * PIPE_DEPTH + 2)]
        self.n = 0
        self.inflight = []

    def full(self):
        return len(self.inflight) >= PIPE_DEPTH

    def submit(self, frame):
        out = self.outs[self.n % len(self.outs)]
        self.n += 1
        self.inflight.append(self.net.submit_u8(frame, out=out, tile_size=self.tile, border=TILE_BORDER if self.tile else 0))

    def collect(self):
        return self.net.collect_u8(self.inflight.pop(0))


class DenoiseStage:
    """`-m n=K` (apply_denoise, upscale/upscale_processing.py:350-354) as a stage of a lane: cv2.fastNlMeansDenoisingColored's
    arithmetic on the lane's GPU (include/uva.h uva_denoise_u8), host frame in, host frame out like the nets' stages -- the call
    takes about a millisecond per 1080p frame and releases the interpreter lock, the nets behind it keep their frames in flight."""

    class _Scale1:
        scale = 1

    def __init__(self, gpu, strength, h, w, alloc):
        self.gpu, self.strength = gpu, float(strength)
        self.net = self._Scale1()
        self.outs = [alloc((h, w, 3)) for _ in range(2 * PIPE_DEPTH + 2)]
        self.n = 0
        self.inflight = []

    def full(self):
        return len(self.inflight) >= 1

    def submit(self, frame):
        from . import _lib
        out = self.outs[self.n % len(self.outs)]
        self.n += 1
        h, w, _ = frame.shape
        _lib.check(_lib.load().uva_denoise_u8(self.gpu, frame.ctypes.data, h, w, frame.strides[0], out.ctypes.data, out.strides[0],
                                              self.strength, self.strength))
        self.inflight.append(out)

    def collect(self):
        return self.inflight.pop(0)


class Lane:
    """The chain of nets of ONE -g entry (one or two Stages on one GPU).  Frames go in with submit() and come out,
    in the order they went in, with pop()."""

    def __init__(self, nets_tiles, h, w, alloc):
        self.stages = []
        for net, tile in nets_tiles:
            if isinstance(net, tuple):         # ("denoise", gpu, K): the `-m n=K` stage
                self.stages.append(DenoiseStage(net[1], net[2], h, w, alloc))
                continue
            self.stages.append(Stage(net, h, w, tile, alloc))
            h, w = h * net.scale, w * net.scale
        self.count = 0                         # frames inside

    def _shift(self):
        """Finished frames move on to the next stage wherever that stage has room and this one is full."""
        for k in range(len(self.stages) - 2, -1, -1):
            while self.stages[k].full() and not self.stages[k + 1].full():
                self.stages[k + 1].submit(self.stages[k].collect())

    def full(self):
        self._shift()
        return self.stages[0].full()

    def submit(self, frame):
        assert not self.full()
        self.stages[0].submit(frame)
        self.count += 1

    def pop(self):
        """The oldest frame's result (blocks until the GPU is done with it)."""
        assert self.count > 0, "pop() on an empty lane"

        def ensure(k):
            assert k >= 0, "no frame in flight in any stage"
            if self.stages[k].inflight:
                return
            ensure(k - 1)
            self.stages[k].submit(self.stages[k - 1].collect())
        last = len(self.stages) - 1
        ensure(last)
        self.count -= 1
        return self.stages[last].collect()


def _regular_fd(f):
    """the descriptor behind f if it is a regular file we may write to with os.pwrite, else None"""
    import stat
    try:
        fd = f.fileno()
        return fd if stat.S_ISREG(os.fstat(fd).st_mode) else None
    except (AttributeError, OSError, ValueError):
        return None


def stream(fin, fout, h, w, nets_tiles, alloc=None, max_frames=None, write_threads=1):
    """Reads u8 [h][w][3] frames from fin until EOF, pushes each through the nets in order, writes the
    results to fout.  nets_tiles: list of (Net, tile_size) -- one GPU worker -- or a list of such lists,
    one per `-g` entry (duplicates allowed, as in the reference's worker list, README.md:45-61): ONE reader
    deals the frames out round-robin (the partition of upscale_processing.py:565-598, frames being
    independent units), every entry keeps up to PIPE_DEPTH frames in flight per net, and ONE writer puts
    the results out in frame order.  write_threads > 1 and fout a regular file: every result frame is cut into that many pieces
    written with os.pwrite by a small pool (one thread copies ~5 GB/s into the page cache: a 4K frame every 5 ms, half of what
    one GPU delivers).  Returns the number of frames written."""
    alloc = alloc or ncnn.pinned_empty
    lanes_spec = nets_tiles if nets_tiles and isinstance(nets_tiles[0], list) else [nets_tiles]
    lanes = [Lane(spec, h, w, alloc) for spec in lanes_spec]
    nl = len(lanes)
    # an input buffer is free again once its frame has left its lane's first stage: at most PIPE_DEPTH per lane are
    # there, plus the one being read into
    ins = [alloc((h, w, 3)) for _ in range(nl * PIPE_DEPTH + 2)]

    # writer thread: the blocking write of frame i overlaps the read of frame i+k and the GPUs
    wq = queue.Queue(maxsize=2)
    werr = []

    fd = _regular_fd(fout) if write_threads > 1 else None
    pool = None
    if fd is not None:
        from concurrent.futures import ThreadPoolExecutor
        fout.flush()
        pos0 = fout.tell()
        pool = ThreadPoolExecutor(max_workers=write_threads)

    def pwrite_all(view, pos):
        while len(view):
            k = os.pwrite(fd, view, pos)
            view, pos = view[k:], pos + k

    sink = PipeSink(fout)

    def writer():
        pos = pos0 if fd is not None else 0
        try:
            while True:
                item = wq.get()
                if item is None:
                    if fd is not None:
                        fout.seek(pos)
                    return
                if fd is None:
                    sink.write(item)
                    continue
                view = memoryview(item).cast("B")
                n = len(view)
                per = -(-n // write_threads)
                piece = -(-per // (1 << 20)) * (1 << 20)              # whole MiB per writer
                jobs = [pool.submit(pwrite_all, view[o:o + piece], pos + o) for o in range(0, n, piece)]
                for j in jobs:
                    j.result()
                pos += n
        except Exception as e:  # noqa: BLE001
            werr.append(e)
            while wq.get() is not None:
                pass

    wt = threading.Thread(target=writer, daemon=True)
    wt.start()
    written = 0

    def pop_next():
        """frame number `written` comes out of its lane and goes to the writer"""
        nonlocal written
        res = lanes[written % nl].pop()
        if werr:
            raise werr[0]
        wq.put(res)
        written += 1

    n_in = 0
    try:
        while max_frames is None or n_in < max_frames:
            buf = ins[n_in % len(ins)]
            if not read_exact(fin, memoryview(buf).cast("B")):
                break
            lane = lanes[n_in % nl]
            while lane.full():                 # frames leave in global order: at most nl pops free a slot of this lane
                pop_next()
            lane.submit(buf)
            n_in += 1
        while written < n_in:
            pop_next()
    finally:
        wq.put(None)
        wt.join()
        if pool is not None:
            pool.shutdown()
    if werr:
        raise werr[0]
    fout.flush()
    sink.drain()
    return written


def filesystem_type(path):
    """fstype of the mount `path` lives on (/proc/self/mountinfo, longest mount point that is a prefix); None if unknown"""
    try:
        real = os.path.realpath(path)
        best, kind = "", None
        for line in open("/proc/self/mountinfo"):
            left, _, right = line.partition(" - ")
            mp = left.split()[4].replace("\\040", " ")
            if (real == mp or real.startswith(mp.rstrip("/") + "/")) and len(mp) >= len(best):
                best, kind = mp, right.split()[0]
        return kind
    except (OSError, IndexError):
        return None


class MappedSegment:
    """The writer's end of ONE worker's byte range of a shared output file: a shared mapping of just that range, filled frame by
    frame with plain memory copies (numpy releases the interpreter lock for them).  Why not write(): every buffered write() /
    pwrite() on a file holds its inode lock for the whole copy -- ext4, xfs and tmpfs alike -- so N writers into ONE file
    proceed one at a time however disjoint their offsets (measured: four workers SLOWER than one,
    profiles/r04_final_rawvideo_bench.txt); stores through a mapping fault their pages in independently.  The pages of a
    frame are asked for in one call (MADV_POPULATE_WRITE, Linux 5.14+) instead of 6 075 faults where the kernel knows it."""
    _POPULATE_WRITE = getattr(mmap, "MADV_POPULATE_WRITE", 23)

    def __init__(self, f, offset, nbytes):
        self._f = f
        gran = mmap.ALLOCATIONGRANULARITY
        base = offset - offset % gran
        self._pos = offset - base
        try:
            self._mm = mmap.mmap(f.fileno(), self._pos + nbytes, access=mmap.ACCESS_WRITE, offset=base)
        except Exception:
            f.close()                            # (the handle was handed over: nobody else closes it)
            raise
        self._arr = np.frombuffer(self._mm, np.uint8)
        self._populate = hasattr(self._mm, "madvise")

    def write(self, view):
        src = np.frombuffer(view, np.uint8)
        n, p = src.size, self._pos
        if p + n > self._arr.size:
            raise ValueError("write past the end of the segment")
        if self._populate:
            a = p - p % mmap.PAGESIZE
            try:
                self._mm.madvise(self._POPULATE_WRITE, a, p + n - a)
            except (OSError, ValueError):
                self._populate = False          # an older kernel: plain page faults do the same, one page at a time
        np.copyto(self._arr[p:p + n], src)
        self._pos = p + n
        return n

    def flush(self):
        pass                                    # (the page cache has the bytes; durability is the caller's fsync, as with write())

    def close(self):
        self._arr = None
        self._mm.close()
        self._f.close()

    def __enter__(self):
        return self

    def __exit__(self, *exc):
        self.close()


def stream_segments(in_path, out_path, h, w, lanes_spec, scale_total, max_frames=None, alloc=None, opener=open, mapped=None,
                    write_threads=1):
    """The same frames -> the same bytes as stream(), for regular FILES, with NO shared serial copy: one contiguous segment of
    frames per `-g` entry (the reference's batches, upscale/upscale_processing.py:923-948, are such segments), and every entry
    runs its own stream() -- its own reader, its own pipelined chain of nets, its own writer thread -- on its own file handles.
      in_path   one file: cut into len(lanes_spec) segments, each read at its offset;  a list: one input file per entry
      out_path  one file: sized first, then every entry writes at the offset its results belong to, on its own handle (`write_threads`
                positional writers per entry); `mapped=True` / UVA_RAW_MMAP=1: through a shared mapping of its byte range instead
                (MappedSegment; space checked with statvfs, blocks reserved with posix_fallocate where that is cheap -- measured
                slower than write() on tmpfs and on overlayfs, kept as an option);  a list: one output file per entry
    One reader and one writer thread copy ~10 GB/s each, two or three GPUs' worth of 1080p -> 4K frames; N independent pairs
    scale with the entries.  A worker that fails takes the output with it: the file(s) this call created are removed (a
    full-size file of zeros and holes is worse than none).  Returns the number of frames written."""
    fb_in, fb_out = h * w * 3, h * scale_total * w * scale_total * 3
    nl = len(lanes_spec)
    ins = list(in_path) if isinstance(in_path, (list, tuple)) else None
    outs = list(out_path) if isinstance(out_path, (list, tuple)) else None
    if (ins is not None and len(ins) != nl) or (outs is not None and len(outs) != nl):
        raise ValueError("one input / output file per -g entry: %d entries" % nl)
    if mapped is None:
        mapped = os.environ.get("UVA_RAW_MMAP", "0") == "1"
    for i in (ins if ins is not None else [in_path]):
        for o in (outs if outs is not None else [out_path]):
            if os.path.exists(o) and os.path.samefile(i, o):
                raise ValueError("%s is input and output at once: the output is sized (truncated) before anything is read" % o)

    def frames_of(path):
        size = os.path.getsize(path)
        if size % fb_in:
            raise EOFError("%s ends inside a frame (%d bytes, frames of %d)" % (path, size, fb_in))
        return size // fb_in
    if ins is None:
        total = frames_of(in_path)
        if max_frames is not None:
            total = min(total, max_frames)
        bounds = [total * k // nl for k in range(nl + 1)]
        counts = [bounds[k + 1] - bounds[k] for k in range(nl)]
        in_first = bounds[:nl]
    else:
        counts, left = [], max_frames
        for path in ins:                         # --frames counts through the files in order
            c = frames_of(path) if left is None else min(frames_of(path), left)
            left = None if left is None else left - c
            counts.append(c)
        in_first = [0] * nl
        total = sum(counts)
    out_first = [0] * nl if outs is not None else [sum(counts[:k]) for k in range(nl)]
    created = [o for o in (outs if outs is not None else [out_path]) if not os.path.exists(o)]
    if outs is None:
        with opener(out_path, "wb") as f:       # the output exists at its full size before anybody writes into it
            f.truncate(total * fb_out)
            if mapped and total:
                # stores into a hole of a full file system end in SIGBUS, not in ENOSPC: make sure of the space first
                st = os.statvfs(out_path)
                if st.f_bavail * st.f_frsize < total * fb_out:
                    f.truncate(0)
                    raise OSError(28, "%s: %d bytes of results do not fit the file system (%d free)"
                                  % (out_path, total * fb_out, st.f_bavail * st.f_frsize))
                if filesystem_type(out_path) not in ("tmpfs", "ramfs", None):
                    try:
                        os.posix_fallocate(f.fileno(), 0, total * fb_out)
                    except OSError:
                        pass                     # (a file system without fallocate: the check above stands)
    done, errs = [0] * nl, []

    def work(k):
        try:
            if counts[k] == 0:
                if outs is not None:
                    opener(outs[k], "wb").close()
                return
            with opener(in_path if ins is None else ins[k], "rb") as fin:
                fin.seek(in_first[k] * fb_in)
                if outs is None and mapped:
                    with MappedSegment(opener(out_path, "r+b"), out_first[k] * fb_out, counts[k] * fb_out) as fout:
                        done[k] = stream(fin, fout, h, w, lanes_spec[k], alloc=alloc, max_frames=counts[k])
                else:
                    with opener(out_path if outs is None else outs[k], "r+b" if outs is None else "wb") as fout:
                        fout.seek(out_first[k] * fb_out)
                        done[k] = stream(fin, fout, h, w, lanes_spec[k], alloc=alloc, max_frames=counts[k], write_threads=write_threads)
        except Exception as e:  # noqa: BLE001
            errs.append(e)
    threads = [threading.Thread(target=work, args=(k,), daemon=True) for k in range(nl)]
    for t in threads:
        t.start()
    for t in threads:
        t.join()
    if errs:
        for o in created:
            try:
                os.remove(o)
            except OSError:
                pass
        raise errs[0]
    return sum(done)


def grow_pipe(f):
    """A frame is 6-100 MB and a pipe holds 64 KiB by default: ask for the largest buffer the system gives an
    unprivileged process (/proc/sys/fs/pipe-max-size, usually 1 MiB) -- 16 times fewer wake-ups per frame."""
    import fcntl
    import stat
    try:
        if not stat.S_ISFIFO(os.fstat(f.fileno()).st_mode):
            return
        want = int(open("/proc/sys/fs/pipe-max-size").read())
        fcntl.fcntl(f.fileno(), getattr(fcntl, "F_SETPIPE_SZ", 1031), want)
    except (OSError, ValueError, AttributeError):
        pass


class PipeSink:
    """The writer's end of a PIPE (stdout into ffmpeg): result frames go in with vmsplice(2) -- the pipe then refers to the pages
    of our page-locked result buffer and the only copy left is the reader's -- instead of write(2), which allocates a pipe page
    and copies into it for every 4 KiB (one thread manages ~6 GB/s of that: 250 4K frames a second, half of what one GPU
    delivers; profiles/r05_final_rawvideo_bench.txt).  No SPLICE_F_GIFT: the pages stay ours, so a buffer must not be rewritten
    while the pipe still refers to it -- vmsplice returns once the LAST piece of a frame is in the pipe, whose capacity is far
    below one frame, so when frame k + 1 has been handed over frame k has been read completely; the result rings hold eight
    buffers (Stage.outs).  Falls back to write() for good the first time the kernel refuses (memory it cannot take
    references on, a pipe that went away is an error as before)."""

    class _IoVec(ctypes.Structure):
        _fields_ = [("base", ctypes.c_void_p), ("len", ctypes.c_size_t)]

    def __init__(self, f):
        import stat
        self._f = f
        self._fd = None
        self._cap = None            # the pipe's capacity, asked for at the first frame (grow_pipe may come after the constructor)
        self.spliced = 0            # frames that went through vmsplice (tests, the bench's log line)
        try:
            fd = f.fileno()
            if stat.S_ISFIFO(os.fstat(fd).st_mode) and os.environ.get("UVA_RAW_VMSPLICE", "1") != "0":
                self._libc = ctypes.CDLL(None, use_errno=True)
                self._libc.vmsplice.restype = ctypes.c_ssize_t
                self._fd = fd
        except (AttributeError, OSError, ValueError):
            self._fd = None

    def write(self, arr):
        """arr: a C-contiguous numpy array (a result buffer of a Stage)"""
        if self._fd is not None and self._cap is None:
            import fcntl
            try:
                self._cap = fcntl.fcntl(self._fd, getattr(fcntl, "F_GETPIPE_SZ", 1032))
            except OSError:
                self._cap = 1 << 30
        if self._fd is None or arr.nbytes < self._cap:        # (frames smaller than the pipe: several could sit in it, still referred to)
            self._f.write(memoryview(arr).cast("B"))
            return
        self._f.flush()             # (nothing of ours is buffered in front of the spliced bytes)
        addr, n, off = arr.ctypes.data, arr.nbytes, 0
        while off < n:
            iov = self._IoVec(addr + off, n - off)
            k = self._libc.vmsplice(self._fd, ctypes.byref(iov), 1, 0)
            if k < 0:
                err = ctypes.get_errno()
                if err == 4:        # EINTR
                    continue
                if off == 0 and err in (14, 22, 38):      # EFAULT / EINVAL / ENOSYS: this memory or this kernel cannot
                    self._fd = None
                    self._f.write(memoryview(arr).cast("B"))
                    return
                raise OSError(err, os.strerror(err))
            off += k
        self.spliced += 1

    def flush(self):
        self._f.flush()

    def drain(self, timeout=60.0):
        """After the LAST frame: wait until the reader has taken every spliced byte out of the pipe (FIONREAD = 0).  The pipe
        refers to our buffers' pages; the caller is about to free them (page-locked memory goes back to the driver's pool), and
        what the reader then found in the pipe's last megabyte would be whatever the pool did with those pages."""
        if self._fd is None or not self.spliced:
            return
        import array
        import fcntl
        import termios
        import time
        t0 = time.monotonic()
        n = array.array("i", [0])
        while time.monotonic() - t0 < timeout:
            try:
                fcntl.ioctl(self._fd, termios.FIONREAD, n, True)
            except OSError:
                return                      # (the reader went away: nothing to protect)
            if n[0] == 0:
                return
            time.sleep(0.001)


def copy_through(fin, fout, h, w, max_frames=None):
    """`-s 1` without `-m a`: the reference renames the frames, nothing is computed (:924-929)."""
    buf = bytearray(h * w * 3)
    n = 0
    while max_frames is None or n < max_frames:
        if not read_exact(fin, memoryview(buf)):
            break
        fout.write(buf)
        n += 1
    fout.flush()
    return n


def main(argv=None):
    ap = argparse.ArgumentParser(description=__doc__.split("\n\n")[0])
    ap.add_argument("-i", "--input", default="-", help="bgr24 rawvideo file, '-' = stdin; a,b,...: one segment file per -g entry")
    ap.add_argument("-o", "--output", default="-", help="bgr24 rawvideo file, '-' = stdout; x,y,...: one output file per -g entry "
                                                        "(a single input is cut into segments, segment k goes to the k-th file)")
    ap.add_argument("-W", "--width", type=int, required=True)
    ap.add_argument("-H", "--height", type=int, required=True)
    ap.add_argument("-s", "--scale", type=int, default=2, choices=[1, 2, 4])
    ap.add_argument("-m", "--models", default="", help="upscale_video.py -m: a = 1x HurrDeblur pass first, n=K = film-grain denoise "
                                                       "(K 1..30) before everything, r = the x_Valar_v1 model (scale 4, needs its .bin)")
    ap.add_argument("-g", "--gpu", default="0",
                    help="HIP ordinals, one worker per entry, e.g. 0,1,2,3 or 0,0,1 (upscale_video.py -g; default 0)")
    ap.add_argument("--tile", type=int, default=TILE_SIZE, help="reference tile size of the final pass (960); 0 = whole frame")
    ap.add_argument("--frames", type=int, default=None, help="stop after this many frames")
    ap.add_argument("--write-threads", type=int, default=1,
                    help="positional writers per worker for a regular output file (default 1: more gained 10 %% on a disk-backed file "
                         "system and lost 40 %% on tmpfs)")
    ap.add_argument("--round-robin", action="store_true",
                    help="file to file with several -g entries: deal the frames out one by one through ONE reader and ONE writer "
                         "(what pipes get) instead of one contiguous segment of frames, reader and writer per entry")
    ap.add_argument("--model-path", default=os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "models"))
    a = ap.parse_args(argv)
    if a.width <= 0 or a.height <= 0:
        ap.error("frame size must be positive")
    # upscale_video.py -m: a (anime pass), n=K (film-grain denoise, K = 1..30, :782-789), r (the x_Valar_v1 model instead of
    # x_Compact_Pretrain, :913-916); the reference runs them in the order n, a, upscale (:880-920) whatever the order given
    models = [m for m in a.models.split(",") if m]
    denoise = None
    for m in models:
        if m.startswith("n="):
            try:
                denoise = int(m[2:])
            except ValueError:
                ap.error("-m n=K takes an integer K")
            if not 1 <= denoise <= 30:
                ap.error("-m n=K: K must be between 1 and 30")
        elif m not in ("a", "r"):
            ap.error("unknown model option %r (a, n=K, r)" % m)
    final_stem = MODEL_FILES[a.scale]
    if "r" in models:
        if a.scale != 4:
            ap.error("-m r: the reference has x_Valar_v1 at scale 4 only (models/4x_Valar_v1.param)")
        final_stem = "4x_Valar_v1"
        if not os.path.exists(os.path.join(a.model_path, final_stem + ".bin")):
            ap.error("-m r: %s.bin is not in %s (a missing blob upstream: supply it with --model-path)" % (final_stem, a.model_path))
    # upscale_video.py: `-m a` runs the 1x HurrDeblur pass (process_model, :888-909), `-s 2|4` the Compact net
    # (upscale_frames, :930-944), and `-s 1` performs NO network pass of its own: its frames are only renamed
    # (:924-929).  So `-s 1` alone copies frames through, `-s 1 -m a` is the HurrDeblur pass alone.
    try:
        gpus = [int(tok) for tok in str(a.gpu).split(",") if tok != ""] or [0]
    except ValueError:
        ap.error("-g takes a comma-separated list of GPU ordinals")
    nets = []                     # one chain of nets per -g entry
    for gpu in gpus:
        chain = []
        if denoise is not None:
            chain.append((("denoise", gpu, denoise), 0))
        if "a" in models:
            chain.append((load_net(MODEL_FILES[1], gpu, a.model_path), 0))          # apply_model: whole frame
        if a.scale != 1:
            chain.append((load_net(final_stem, gpu, a.model_path), a.tile))
        if chain:
            nets.append(chain)
    # file -> file with several workers: one contiguous segment of frames per worker, each with its own reader and writer.
    # "a,b,..." is a LIST only where a list means something -- more than one -g entry -- and only if no file of exactly that
    # name exists (a path may contain a comma)
    def as_list(arg):
        if len(gpus) > 1 and "," in arg and not os.path.exists(arg):
            return arg.split(",")
        return [arg]
    ins, outs = as_list(a.input), as_list(a.output)
    if len(ins) > 1 or len(outs) > 1:
        if not nets:
            ap.error("-i / -o lists give every -g entry's NETWORK PASS a file of its own; `-s 1` without -m a / n=K only copies "
                     "frames through (upscale_processing.py:924-929): one input, one output")
        if (len(ins) > 1 and len(ins) != len(nets)) or (len(outs) > 1 and len(outs) != len(nets)) or "-" in ins + outs:
            ap.error("-i / -o lists: one regular file per -g entry (%d entries)" % len(nets))
    for i in ins:
        for o in outs:
            if i != "-" and o != "-" and os.path.exists(i) and os.path.exists(o) and os.path.samefile(i, o):
                ap.error("%s is input and output at once" % o)
    wthreads = max(1, a.write_threads)
    regular = all(os.path.isfile(f) for f in ins) and all(not os.path.exists(f) or os.path.isfile(f) for f in outs)
    # ONE output file on tmpfs: every write into it allocates its pages under the inode's lock, so several workers writing their
    # segments into it are SLOWER than one (246.7 -> 208.3 -> 162.4 frames/s at 1 / 4 / 8 workers, profiles/r05_final_rawvideo_bench.txt;
    # a disk-backed file system scales: 201 -> 431 -> 614).  There the workers feed ONE writer thread instead (the pipe route's
    # shape: one reader, frames dealt round-robin, results written in order) -- never slower than one worker.  `-o a,b,..` (a file
    # per worker) is the way past the lock and keeps its segments.
    one_file_on_tmpfs = (len(nets) > 1 and len(outs) == 1 and a.output != "-" and
                         filesystem_type(os.path.dirname(os.path.abspath(a.output)) or ".") in ("tmpfs", "ramfs") and
                         os.environ.get("UVA_RAW_TMPFS_SEGMENTS") != "1")
    if nets and (len(ins) > 1 or len(outs) > 1 or (len(nets) > 1 and not a.round_robin and not one_file_on_tmpfs and
                                                   a.input != "-" and a.output != "-" and regular)):
        if not regular:
            ap.error("-i / -o lists take regular files")
        scale_total = 1
        for net, _ in nets[0]:
            scale_total *= 1 if isinstance(net, tuple) else net.scale
        n = stream_segments(ins if len(ins) > 1 else ins[0], outs if len(outs) > 1 else outs[0], a.height, a.width, nets, scale_total,
                            max_frames=a.frames, write_threads=wthreads)
        print("%d frames" % n, file=sys.stderr)
        ncnn.destroy_gpu_instance()
        return 0
    fin = sys.stdin.buffer if a.input == "-" else open(a.input, "rb")
    created = a.output != "-" and not os.path.exists(a.output)
    fout = sys.stdout.buffer if a.output == "-" else open(a.output, "wb")
    for f in (fin, fout):
        grow_pipe(f)
    ok = False
    try:
        if nets:
            n = stream(fin, fout, a.height, a.width, nets, max_frames=a.frames, write_threads=wthreads)
        else:
            n = copy_through(fin, fout, a.height, a.width, a.frames)
        ok = True
    finally:
        if fin is not sys.stdin.buffer:
            fin.close()
        if fout is not sys.stdout.buffer:
            fout.close()
            if not ok and created and os.path.isfile(a.output):
                os.remove(a.output)                # half a stream under the name of a whole one helps nobody
    print("%d frames" % n, file=sys.stderr)
    ncnn.destroy_gpu_instance()
    return 0


if __name__ == "__main__":
    sys.exit(main())
