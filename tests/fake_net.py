"""Stand-in net factory for the CPU tests of the frame workers (frame_pool.FramePool takes the factory as
"module:function"): nearest-neighbour upscale + 1 with the Net.submit_u8 / collect_u8 contract.  The model
file name selects a behaviour; pixel [0, 0, 0] of a frame carries its frame number."""
import os

import numpy as np


class FakeNet:
    def __init__(self, scale, die_at=None, fail_at=None, journal=None):
        self.scale, self.die_at, self.fail_at, self.journal = max(1, scale), die_at, fail_at, journal
        self.live = 0

    def submit_u8(self, img, out=None, tile_size=0, border=0):
        assert self.live < 3, "more than 3 frames in flight"
        frame = int(img[0, 0, 0])
        if self.die_at is not None and frame == self.die_at:
            os._exit(17)                                   # a crashed worker: no clean-up, no report
        if self.fail_at is not None and frame == self.fail_at:
            raise RuntimeError("extract failed on frame %d" % frame)
        self.live += 1
        return (img, out, frame, tile_size, border)

    def collect_u8(self, t):
        img, out, frame, tile_size, border = t
        self.live -= 1
        out[...] = np.repeat(np.repeat(img, self.scale, 0), self.scale, 1) + 1
        if self.journal:
            with open(self.journal, "a") as f:
                f.write("%d %d %d %d\n" % (os.getpid(), frame, tile_size, border))
        return out


def make(model_path, model_file, scale, gpu):
    """model_path doubles as the journal directory"""
    journal = os.path.join(model_path, "journal.txt")
    with open(os.path.join(model_path, "loads.txt"), "a") as f:
        f.write("%d %s %d %d\n" % (os.getpid(), model_file, scale, gpu))
    if model_file.startswith("x_bad"):
        raise RuntimeError("Unable to load model " + model_file)
    die_at = int(model_file.split("die")[1]) if "die" in model_file else None
    fail_at = int(model_file.split("fail")[1]) if "fail" in model_file else None
    return FakeNet(scale, die_at, fail_at, journal)
