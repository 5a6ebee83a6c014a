"""rawvideo streamer (upscale_video_amd/rawvideo.py): host logic on CPU with a stand-in net, the real
engine under -m gpu."""
import io
import os

import numpy as np
import pytest

from upscale_video_amd import rawvideo


class FakeNet:
    """Stand-in with the Net.submit_u8 / collect_u8 contract: nearest-neighbour upscale + 1, finished
    only at collect time (so that buffer-reuse mistakes show up)."""

    def __init__(self, scale):
        self.scale = scale
        self.live = 0

    def submit_u8(self, img, out=None, tile_size=0, border=0):
        assert self.live < 3, "more than 3 frames in flight"
        self.live += 1
        return (img, out)                      # keeps a VIEW of the input: it must still be intact at collect

    def collect_u8(self, t):
        img, out = t
        self.live -= 1
        out[...] = np.repeat(np.repeat(img, self.scale, 0), self.scale, 1) + 1
        return out


def _frames(n, h, w):
    rng = np.random.default_rng(7)
    return [rng.integers(0, 200, (h, w, 3), dtype=np.uint8) for _ in range(n)]


@pytest.mark.parametrize("nframes", [0, 1, 2, 3, 4, 11, 23])
def test_stream_order_and_buffer_lifetimes_two_stages(nframes):
    h, w = 6, 10
    frames = _frames(nframes, h, w)
    fin = io.BytesIO(b"".join(f.tobytes() for f in frames))
    fout = io.BytesIO()
    alloc = lambda shape: np.empty(shape, np.uint8)   # noqa: E731
    n = rawvideo.stream(fin, fout, h, w, [(FakeNet(1), 0), (FakeNet(2), 960)], alloc=alloc)
    assert n == nframes
    got = np.frombuffer(fout.getvalue(), np.uint8).reshape(nframes, 2 * h, 2 * w, 3)
    for i, f in enumerate(frames):
        want = np.repeat(np.repeat(f + 1, 2, 0), 2, 1) + 1
        assert np.array_equal(got[i], want), i


@pytest.mark.parametrize("nlanes", [2, 3, 8])
@pytest.mark.parametrize("nframes", [0, 1, 5, 16, 37])
def test_several_workers_keep_frame_order_and_buffer_lifetimes(nlanes, nframes):
    """`-g a,b,c`: one chain of nets per entry, frames dealt out round-robin, results written in frame order; every
    net sees at most 3 frames in flight and intact input views (two-stage chains: the `-m a` case)."""
    h, w = 6, 10
    frames = _frames(nframes, h, w)
    fin = io.BytesIO(b"".join(f.tobytes() for f in frames))
    fout = io.BytesIO()
    alloc = lambda shape: np.empty(shape, np.uint8)   # noqa: E731
    lanes = [[(FakeNet(1), 0), (FakeNet(2), 960)] for _ in range(nlanes)]
    assert rawvideo.stream(fin, fout, h, w, lanes, alloc=alloc) == nframes
    got = np.frombuffer(fout.getvalue(), np.uint8).reshape(nframes, 2 * h, 2 * w, 3)
    for i, f in enumerate(frames):
        assert np.array_equal(got[i], np.repeat(np.repeat(f + 1, 2, 0), 2, 1) + 1), i
    assert all(net.live == 0 for lane in lanes for net, _ in lane)


def test_torn_last_frame_is_an_error():
    h, w = 4, 4
    fin = io.BytesIO(_frames(2, h, w)[0].tobytes() + b"\x00" * 10)
    with pytest.raises(EOFError, match="inside a frame"):
        rawvideo.stream(fin, io.BytesIO(), h, w, [(FakeNet(2), 0)], alloc=lambda s: np.empty(s, np.uint8))


def test_max_frames_and_writer_errors():
    h, w = 4, 4
    data = b"".join(f.tobytes() for f in _frames(9, h, w))
    fout = io.BytesIO()
    assert rawvideo.stream(io.BytesIO(data), fout, h, w, [(FakeNet(2), 0)], alloc=lambda s: np.empty(s, np.uint8), max_frames=5) == 5
    assert len(fout.getvalue()) == 5 * 4 * h * w * 3

    class Broken(io.RawIOBase):
        def write(self, b):
            raise BrokenPipeError("downstream closed")

    with pytest.raises(BrokenPipeError):
        rawvideo.stream(io.BytesIO(data), Broken(), h, w, [(FakeNet(2), 0)], alloc=lambda s: np.empty(s, np.uint8))


def test_scale_1_without_the_anime_pass_copies_frames_through(tmp_path):
    """upscale_video.py with -s 1 and no -m a runs no network at all (frames are renamed, :924-929)."""
    h, w = 4, 6
    data = b"".join(f.tobytes() for f in _frames(5, h, w))
    src, dst = tmp_path / "in.bgr24", tmp_path / "out.bgr24"
    src.write_bytes(data)
    assert rawvideo.main(["-i", str(src), "-o", str(dst), "-W", str(w), "-H", str(h), "-s", "1"]) == 0
    assert dst.read_bytes() == data
    assert rawvideo.main(["-i", str(src), "-o", str(dst), "-W", str(w), "-H", str(h), "-s", "1", "--frames", "2"]) == 0
    assert dst.read_bytes() == data[:2 * h * w * 3]


def test_cli_checks_the_model_options(capsys, tmp_path):
    """upscale_video.py -m: a, n=K (K = 1..30), r (x_Valar_v1: scale 4, its .bin a missing blob upstream)"""
    for argv, msg in ((["-m", "r"], "scale 4 only"), (["-m", "r", "-s", "4", "--model-path", str(tmp_path)], "missing blob"),
                      (["-m", "n=0"], "between 1 and 30"), (["-m", "n=x"], "integer"), (["-m", "q"], "unknown model option")):
        with pytest.raises(SystemExit):
            rawvideo.main(["-W", "8", "-H", "8"] + argv)
        assert msg in capsys.readouterr().err, argv


def test_denoise_stage_keeps_its_place_in_a_lane(monkeypatch):
    """`-m n=K,a -s 2`: the denoise stage first (the reference's order), frames in order, buffers alive until consumed"""
    calls = []

    class FakeLib:
        def uva_denoise_u8(self, gpu, src, h, w, ss, dst, ds, hl, hc):
            calls.append((gpu, h, w, hl, hc))
            a = np.ctypeslib.as_array((np.ctypeslib.ctypes.c_uint8 * (h * w * 3)).from_address(src)).reshape(h, w, 3)
            b = np.ctypeslib.as_array((np.ctypeslib.ctypes.c_uint8 * (h * w * 3)).from_address(dst)).reshape(h, w, 3)
            b[...] = 255 - a
            return 0
    from upscale_video_amd import _lib
    monkeypatch.setattr(_lib, "load", lambda: FakeLib())
    monkeypatch.setattr(_lib, "check", lambda rc: None)
    h, w = 6, 10
    frames = _frames(9, h, w)
    fin = io.BytesIO(b"".join(f.tobytes() for f in frames))
    fout = io.BytesIO()
    n = rawvideo.stream(fin, fout, h, w, [(("denoise", 3, 7), 0), (FakeNet(1), 0), (FakeNet(2), 32)], alloc=lambda shape: np.zeros(shape, np.uint8))
    assert n == 9 and calls[0] == (3, h, w, 7.0, 7.0) and len(calls) == 9
    got = np.frombuffer(fout.getvalue(), np.uint8).reshape(9, 2 * h, 2 * w, 3)
    for i, f in enumerate(frames):
        want = np.repeat(np.repeat((255 - f) + 1, 2, 0), 2, 1) + 1
        assert np.array_equal(got[i], want), i


@pytest.mark.gpu
def test_stream_equals_per_frame_calls(tmp_path):
    import torch  # noqa: F401  (its HIP runtime first, see test_gpu_parity.py)
    from conftest import load_net
    from upscale_video_amd import ncnn
    from oracle import uvoracle
    h, w, n = 72, 112, 7
    frames = [uvoracle.synthetic_frame(h, w, seed=50 + i) for i in range(n)]
    src = tmp_path / "in.bgr24"
    src.write_bytes(b"".join(f.tobytes() for f in frames))
    dst = tmp_path / "out.bgr24"
    assert rawvideo.main(["-i", str(src), "-o", str(dst), "-W", str(w), "-H", str(h), "-s", "2", "-m", "a", "--tile", "32"]) == 0
    got = np.frombuffer(dst.read_bytes(), np.uint8).reshape(n, 2 * h, 2 * w, 3)
    pre, net = load_net(ncnn, "1x"), load_net(ncnn, "2x")
    o1, o2 = uvoracle.load_model("1x"), uvoracle.load_model("2x")
    for i, f in enumerate(frames):
        # against the CPU oracle's chain (apply_model -> u8 -> upscale_image with the same 32/10 tiling) ...
        want = o2.upscale_image(o1.apply_model(f), tile_size=32, border=10)
        d = np.abs(got[i].astype(int) - want.astype(int))
        mse = float((d.astype(np.float64) ** 2).mean())
        assert d.max() <= 3 and (mse == 0 or 10 * np.log10(255.0 ** 2 / mse) >= 48), (i, int(d.max()))
        # ... and, bit for bit, against the synchronous per-frame calls of the same engine
        assert np.array_equal(got[i], net.process_u8(pre.process_u8(f), tile_size=32, border=10)), i
    # `-m n=3,a -s 2`: denoise -> anime pass -> 2x in the reference's order (upscale_processing.py:880-920), bit for bit the
    # per-stage calls (the file route's arithmetic: apply_denoise, apply_model, upscale_image) ...
    from upscale_video_amd import upscale_processing as up
    dst3 = tmp_path / "out3.bgr24"
    assert rawvideo.main(["-i", str(src), "-o", str(dst3), "-W", str(w), "-H", str(h), "-s", "2", "-m", "a,n=3", "--tile", "32"]) == 0
    got3 = np.frombuffer(dst3.read_bytes(), np.uint8).reshape(n, 2 * h, 2 * w, 3)
    for i, f in enumerate(frames):
        assert np.array_equal(got3[i], net.process_u8(pre.process_u8(up.denoise_u8(f, 3, device=0)), tile_size=32, border=10)), i
    # ... and the device-resident form of the same chain (uva_denoise_u8_device queued in front of the 1x net)
    d_in, d_mid = torch.from_numpy(frames[0]).cuda(), torch.empty((h, w, 3), dtype=torch.uint8, device="cuda")
    d_mid2, d_out = torch.empty_like(d_mid), torch.empty((2 * h, 2 * w, 3), dtype=torch.uint8, device="cuda")
    torch.cuda.synchronize()
    pre.denoise_u8_device(d_in.data_ptr(), h, w, d_mid.data_ptr(), 3)
    pre.process_u8_device(d_mid.data_ptr(), h, w, d_mid2.data_ptr(), tile_size=0)
    net.wait_for(pre)
    net.process_u8_device(d_mid2.data_ptr(), h, w, d_out.data_ptr(), tile_size=32, border=10)
    net.synchronize()
    assert np.array_equal(d_out.cpu().numpy(), got3[0])
    # two workers on the one GPU (`-g 0,0`, the reference's duplicate entries): same bytes, same order
    dst2 = tmp_path / "out2.bgr24"
    assert rawvideo.main(["-i", str(src), "-o", str(dst2), "-W", str(w), "-H", str(h), "-s", "2", "-m", "a", "--tile", "32", "-g", "0,0"]) == 0
    assert dst2.read_bytes() == dst.read_bytes()


@pytest.mark.parametrize("nlanes,nframes", [(8, 37), (8, 5), (3, 10), (2, 0)])
def test_segments_have_their_own_reader_and_writer(tmp_path, nlanes, nframes):
    """file -> file with several workers: one contiguous segment of frames per lane, every lane on its
    own file handles and threads -- the bytes of the one-reader route, and no handle or thread shared between lanes"""
    import threading
    h, w = 6, 10
    frames = _frames(nframes, h, w)
    src, dst = tmp_path / "in.bgr24", tmp_path / "out.bgr24"
    src.write_bytes(b"".join(f.tobytes() for f in frames))
    opened = []

    def opener(path, mode):
        f = open(path, mode)
        opened.append((str(path), mode, threading.current_thread().name, f))
        return f
    lanes = [[(FakeNet(1), 0), (FakeNet(2), 32)] for _ in range(nlanes)]
    n = rawvideo.stream_segments(str(src), str(dst), h, w, lanes, 2, alloc=lambda shape: np.zeros(shape, np.uint8), opener=opener)
    assert n == nframes
    got = np.frombuffer(dst.read_bytes(), np.uint8).reshape(nframes, 2 * h, 2 * w, 3)
    for i, f in enumerate(frames):
        assert np.array_equal(got[i], np.repeat(np.repeat(f + 1, 2, 0), 2, 1) + 1), i
    busy = min(nlanes, nframes)        # lanes that got at least one frame
    readers = [o for o in opened if o[1] == "rb"]
    writers = [o for o in opened if o[1] == "r+b"]
    assert len(readers) == len(writers) == busy
    assert len({o[2] for o in readers}) == busy and len({id(o[3]) for o in readers + writers}) == 2 * busy
    # the same bytes through ONE reader and ONE writer dealing the frames out round-robin
    import io
    fout = io.BytesIO()
    rawvideo.stream(io.BytesIO(src.read_bytes()), fout, h, w, [[(FakeNet(1), 0), (FakeNet(2), 32)] for _ in range(nlanes)],
                    alloc=lambda shape: np.zeros(shape, np.uint8))
    assert fout.getvalue() == dst.read_bytes()


def test_segments_to_and_from_one_file_per_lane(tmp_path):
    """-o x,y,... : segment k of the single input goes to the k-th output file; -i a,b,... : one input file per lane;
    the concatenation is the one-reader route's output either way"""
    import io
    h, w, nlanes, nframes = 6, 10, 4, 11
    frames = _frames(nframes, h, w)
    src = tmp_path / "in.bgr24"
    src.write_bytes(b"".join(f.tobytes() for f in frames))
    mk = lambda: [[(FakeNet(1), 0), (FakeNet(2), 32)] for _ in range(nlanes)]   # noqa: E731
    alloc = lambda shape: np.zeros(shape, np.uint8)                             # noqa: E731
    whole = io.BytesIO()
    rawvideo.stream(io.BytesIO(src.read_bytes()), whole, h, w, mk(), alloc=alloc)
    outs = [str(tmp_path / ("out%d.bgr24" % k)) for k in range(nlanes)]
    assert rawvideo.stream_segments(str(src), outs, h, w, mk(), 2, alloc=alloc) == nframes
    assert b"".join(open(o, "rb").read() for o in outs) == whole.getvalue()
    # the segments as input files of their own, results into one file; --frames counts through the inputs in order
    fb = h * w * 3
    bounds = [nframes * k // nlanes for k in range(nlanes + 1)]
    ins = []
    for k in range(nlanes):
        ins.append(str(tmp_path / ("in%d.bgr24" % k)))
        open(ins[-1], "wb").write(src.read_bytes()[bounds[k] * fb:bounds[k + 1] * fb])
    dst = str(tmp_path / "joined.bgr24")
    assert rawvideo.stream_segments(ins, dst, h, w, mk(), 2, alloc=alloc) == nframes
    assert open(dst, "rb").read() == whole.getvalue()
    assert rawvideo.stream_segments(ins, dst, h, w, mk(), 2, alloc=alloc, max_frames=5) == 5
    assert open(dst, "rb").read() == whole.getvalue()[:5 * fb * 4]
    with pytest.raises(ValueError):
        rawvideo.stream_segments(ins[:2], dst, h, w, mk(), 2, alloc=alloc)


@pytest.mark.parametrize("nlanes,nframes,h,w", [(4, 13, 5, 7), (8, 37, 6, 10), (3, 3, 33, 41), (2, 1, 4, 4)])
def test_one_output_file_through_mappings_equals_seek_and_write(tmp_path, nlanes, nframes, h, w):
    """N workers into ONE output file without taking its inode lock in turn -- every worker copies into a
    shared mapping of its own byte range (MappedSegment).  Frame sizes that are no multiple of a page (segment offsets in the
    middle of a page, two workers' ranges sharing one): the bytes of the seek + write route and of the one-writer route."""
    import io
    frames = _frames(nframes, h, w)
    src = tmp_path / "in.bgr24"
    src.write_bytes(b"".join(f.tobytes() for f in frames))
    mk = lambda: [[(FakeNet(1), 0), (FakeNet(2), 32)] for _ in range(nlanes)]   # noqa: E731
    alloc = lambda shape: np.zeros(shape, np.uint8)                             # noqa: E731
    a, b = str(tmp_path / "mapped.bgr24"), str(tmp_path / "written.bgr24")
    assert rawvideo.stream_segments(str(src), a, h, w, mk(), 2, alloc=alloc, mapped=True) == nframes
    assert rawvideo.stream_segments(str(src), b, h, w, mk(), 2, alloc=alloc, mapped=False) == nframes
    whole = io.BytesIO()
    rawvideo.stream(io.BytesIO(src.read_bytes()), whole, h, w, mk(), alloc=alloc)
    assert open(a, "rb").read() == open(b, "rb").read() == whole.getvalue()
    assert os.path.getsize(a) == nframes * h * w * 3 * 4


def test_mapped_segment_refuses_to_write_past_its_range(tmp_path):
    p = tmp_path / "f.bin"
    p.write_bytes(b"\0" * 10000)
    seg = rawvideo.MappedSegment(open(p, "r+b"), 4097, 100)
    seg.write(memoryview(bytes(range(100))))
    with pytest.raises(ValueError, match="past the end"):
        seg.write(memoryview(b"x"))
    seg.close()
    data = p.read_bytes()
    assert data[4097:4197] == bytes(range(100)) and data[:4097] == b"\0" * 4097 and data[4197:] == b"\0" * (10000 - 4197)


class _FailingNet(FakeNet):
    def collect_u8(self, t):
        raise RuntimeError("device lost")


def test_a_failed_worker_takes_the_output_with_it(tmp_path):
    """a worker that fails used to leave a full-size output of zeros and holes under the name of a result"""
    h, w = 6, 10
    src = tmp_path / "in.bgr24"
    src.write_bytes(b"".join(f.tobytes() for f in _frames(8, h, w)))
    lanes = [[(FakeNet(2), 0)], [(_FailingNet(2), 0)]]
    alloc = lambda shape: np.zeros(shape, np.uint8)                             # noqa: E731
    dst = str(tmp_path / "out.bgr24")
    with pytest.raises(RuntimeError, match="device lost"):
        rawvideo.stream_segments(str(src), dst, h, w, lanes, 2, alloc=alloc)
    assert not os.path.exists(dst)
    outs = [str(tmp_path / "o0"), str(tmp_path / "o1")]
    with pytest.raises(RuntimeError, match="device lost"):
        rawvideo.stream_segments(str(src), outs, h, w, [[(FakeNet(2), 0)], [(_FailingNet(2), 0)]], 2, alloc=alloc)
    assert not any(os.path.exists(o) for o in outs)


def test_input_and_output_must_differ_and_commas_are_lists_only_for_worker_lists(tmp_path, capsys):
    """`-i X -o X` destroyed the input (the output is sized before anything is read); a path with a comma in it
    stopped working when -i / -o became lists; a list with `-s 1` (no network pass) got the message '0 entries'"""
    h, w = 4, 6
    same = tmp_path / "same.bgr24"
    same.write_bytes(b"".join(f.tobytes() for f in _frames(2, h, w)))
    with pytest.raises(SystemExit):
        rawvideo.main(["-i", str(same), "-o", str(same), "-W", str(w), "-H", str(h), "-s", "1"])
    assert "input and output at once" in capsys.readouterr().err and same.stat().st_size == 2 * h * w * 3
    with pytest.raises(ValueError, match="input and output at once"):
        rawvideo.stream_segments(str(same), str(same), h, w, [[(FakeNet(2), 0)], [(FakeNet(2), 0)]], 2,
                                 alloc=lambda s: np.zeros(s, np.uint8))
    assert same.stat().st_size == 2 * h * w * 3
    # one worker: "a,b" is a file name (copy-through needs no GPU)
    comma_in = tmp_path / "take,1.bgr24"
    comma_in.write_bytes(same.read_bytes())
    comma_out = tmp_path / "out,1.bgr24"
    assert rawvideo.main(["-i", str(comma_in), "-o", str(comma_out), "-W", str(w), "-H", str(h), "-s", "1"]) == 0
    assert comma_out.read_bytes() == same.read_bytes()
    # a worker list without a network pass: the message says what the lists are for
    with pytest.raises(SystemExit):
        rawvideo.main(["-i", str(same), "-o", "%s,%s" % (tmp_path / "x", tmp_path / "y"), "-W", str(w), "-H", str(h), "-s", "1", "-g", "0,0"])
    assert "NETWORK PASS" in capsys.readouterr().err


def test_filesystem_type_reads_the_mount_table(tmp_path):
    assert rawvideo.filesystem_type("/dev/shm") in ("tmpfs", None)
    assert rawvideo.filesystem_type(str(tmp_path)) is None or isinstance(rawvideo.filesystem_type(str(tmp_path)), str)


@pytest.mark.parametrize("threads", [2, 3, 4])
def test_positional_writers_give_the_bytes_of_the_single_writer(tmp_path, threads):
    """--write-threads: a result frame cut into whole-MiB pieces written with os.pwrite by a pool (regular files only); the bytes
    and the file position behind the stream are those of plain write() calls, also in the middle of a file (a segment's offset)"""
    import io
    h, w, nframes = 300, 500, 5                      # 450 000 B in, 1.8 MB out per frame: two pieces with 2-4 writers
    frames = _frames(nframes, h, w)
    data = b"".join(f.tobytes() for f in frames)
    alloc = lambda shape: np.zeros(shape, np.uint8)   # noqa: E731
    ref = io.BytesIO()
    rawvideo.stream(io.BytesIO(data), ref, h, w, [(FakeNet(2), 0)], alloc=alloc)
    p = tmp_path / "out.bgr24"
    with open(p, "wb") as f:
        f.write(b"HEAD" * 3)                         # the stream starts at the handle's position, not at 0
        n = rawvideo.stream(io.BytesIO(data), f, h, w, [(FakeNet(2), 0)], alloc=alloc, write_threads=threads)
        assert n == nframes and f.tell() == 12 + len(ref.getvalue())
        f.write(b"TAIL")
    got = p.read_bytes()
    assert got[:12] == b"HEAD" * 3 and got[12:-4] == ref.getvalue() and got[-4:] == b"TAIL"
    # a pipe-like object (no descriptor): the single writer, silently
    out = io.BytesIO()
    assert rawvideo.stream(io.BytesIO(data), out, h, w, [(FakeNet(2), 0)], alloc=alloc, write_threads=threads) == nframes
    assert out.getvalue() == ref.getvalue()
    # segments with positional writers: the same file as without
    src = tmp_path / "in.bgr24"
    src.write_bytes(data)
    a, b = str(tmp_path / "a"), str(tmp_path / "b")
    mk = lambda: [[(FakeNet(2), 0)] for _ in range(2)]   # noqa: E731
    rawvideo.stream_segments(str(src), a, h, w, mk(), 2, alloc=alloc, write_threads=threads)
    rawvideo.stream_segments(str(src), b, h, w, mk(), 2, alloc=alloc, write_threads=1)
    assert open(a, "rb").read() == open(b, "rb").read() == ref.getvalue()


@pytest.mark.parametrize("splice", ["1", "0"])
def test_results_into_a_pipe_by_vmsplice_are_the_bytes_of_write(splice, monkeypatch):
    """stdout into ffmpeg is a pipe: result frames larger than the pipe's capacity are handed over with vmsplice (the pipe
    refers to the result buffer's pages, no copy on our side) -- a SLOW reader must still see every frame intact and in order
    although the eight result buffers are rewritten while it reads, and UVA_RAW_VMSPLICE=0 (plain write) gives the same bytes"""
    import threading
    import time
    monkeypatch.setenv("UVA_RAW_VMSPLICE", splice)
    h, w, nframes = 120, 160, 30                    # results 240 x 320 x 3 = 230 KB > the pipe's 64 KB
    frames = _frames(nframes, h, w)
    r, wfd = os.pipe()
    got = []

    def reader():
        with os.fdopen(r, "rb") as f:
            while True:
                b = f.read(2 * h * 2 * w * 3)
                if not b:
                    return
                got.append(b)
                time.sleep(0.003)                   # slower than the producer: buffers come round while frames wait in the pipe
    t = threading.Thread(target=reader)
    t.start()
    with os.fdopen(wfd, "wb") as fout:
        sink_probe = rawvideo.PipeSink(fout)
        assert (sink_probe._fd is not None) == (splice == "1")
        n = rawvideo.stream(io.BytesIO(b"".join(f.tobytes() for f in frames)), fout, h, w, [(FakeNet(2), 0)],
                            alloc=lambda s: np.empty(s, np.uint8))
    t.join()
    assert n == nframes and len(got) == nframes
    for i, f in enumerate(frames):
        assert got[i] == (np.repeat(np.repeat(f, 2, 0), 2, 1) + 1).tobytes(), i


def test_frames_smaller_than_the_pipe_are_written_not_spliced():
    r, wfd = os.pipe()
    with os.fdopen(wfd, "wb") as fout, os.fdopen(r, "rb") as fin:
        sink = rawvideo.PipeSink(fout)
        a = np.arange(300, dtype=np.uint8)
        sink.write(a)
        sink.flush()
        assert sink.spliced == 0 and fin.read(300) == a.tobytes()


def test_the_pipe_is_drained_before_the_buffers_go(monkeypatch):
    """stream() returns only when the reader has taken the last spliced bytes (the pipe refers to the result buffers' pages)"""
    import threading
    import time
    monkeypatch.setenv("UVA_RAW_VMSPLICE", "1")
    h, w, nframes = 120, 160, 4
    frames = _frames(nframes, h, w)
    r, wfd = os.pipe()
    state = {"returned_at": None, "read_done_at": None}
    got = []

    def reader():
        time.sleep(0.3)                                     # the reader starts late: the last frame's tail sits in the pipe
        with os.fdopen(r, "rb") as f:
            while True:
                b = f.read(2 * h * 2 * w * 3)
                if not b:
                    break
                got.append(b)
                if len(got) == nframes:
                    state["read_done_at"] = time.monotonic()
    t = threading.Thread(target=reader)
    t.start()
    with os.fdopen(wfd, "wb") as fout:
        n = rawvideo.stream(io.BytesIO(b"".join(f.tobytes() for f in frames)), fout, h, w, [(FakeNet(2), 0)],
                            alloc=lambda s: np.empty(s, np.uint8))
        state["returned_at"] = time.monotonic()
    t.join()
    assert n == nframes and len(got) == nframes
    assert state["returned_at"] >= state["read_done_at"] - 0.05, state      # not before the reader had everything
