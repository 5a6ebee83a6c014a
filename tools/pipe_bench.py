"""What ONE pipe carries: a producer writes 4K bgr24 frames (24.9 MB) from a ring of buffers into its stdout, `cat > /dev/null`
(or --reader splice: a reader that splices into /dev/null) takes them.  Modes: write() / vmsplice(SPLICE_F_GIFT is NOT used: the
pages stay ours, the ring is long enough that a buffer is rewritten only after the pipe has drained it), pipe capacity 64 KB
(the default), 1 MB (the unprivileged maximum), 16 MB (root).  No GPU needed.

    python tools/pipe_bench.py            # the table
"""
import ctypes
import fcntl
import mmap
import os
import subprocess
import sys
import time

F_SETPIPE_SZ, F_GETPIPE_SZ = 1031, 1032
FRAME = 2160 * 3840 * 3


class IoVec(ctypes.Structure):
    _fields_ = [("base", ctypes.c_void_p), ("len", ctypes.c_size_t)]


def producer(mode, pipe_sz, frames):
    fd = 1
    if pipe_sz:
        try:
            fcntl.fcntl(fd, F_SETPIPE_SZ, pipe_sz)
        except OSError as e:
            print("F_SETPIPE_SZ %d: %s" % (pipe_sz, e), file=sys.stderr)
    got = fcntl.fcntl(fd, F_GETPIPE_SZ)
    ring = [mmap.mmap(-1, FRAME) for _ in range(8)]
    for m in ring:
        m.write(b"\x55" * FRAME)
    libc = ctypes.CDLL(None, use_errno=True)
    t0 = time.perf_counter()
    for i in range(frames):
        m = ring[i % len(ring)]
        if mode == "write":
            view = memoryview(m)
            off = 0
            while off < FRAME:
                off += os.write(fd, view[off:])
        else:
            addr = ctypes.addressof(ctypes.c_char.from_buffer(m))
            off = 0
            while off < FRAME:
                iov = IoVec(addr + off, FRAME - off)
                k = libc.vmsplice(fd, ctypes.byref(iov), 1, 0)
                if k < 0:
                    raise OSError(ctypes.get_errno(), "vmsplice")
                off += k
    dt = time.perf_counter() - t0
    print("%-9s pipe %8d B: %4d frames in %6.2f s = %6.1f frames/s = %5.2f GB/s" % (mode, got, frames, dt, frames / dt, frames * FRAME / dt / 1e9),
          file=sys.stderr)


if __name__ == "__main__":
    if len(sys.argv) > 1 and sys.argv[1] == "producer":
        producer(sys.argv[2], int(sys.argv[3]), int(sys.argv[4]))
        sys.exit(0)
    n = int(sys.argv[1]) if len(sys.argv) > 1 else 200
    for reader in ("cat > /dev/null", "dd bs=1M of=/dev/null status=none"):
        print("reader: " + reader)
        for mode in ("write", "vmsplice"):
            for sz in (0, 1 << 20, 16 << 20):
                subprocess.run("%s %s producer %s %d %d | %s" % (sys.executable, os.path.abspath(__file__), mode, sz, n, reader), shell=True)
