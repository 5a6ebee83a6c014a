"""Device-frame route (uva_net_process_u8_device): asynchronous, ordered only by the net's stream and by uva_net_wait_for /
uva_net_synchronize / the denoise stage's `after`.  Frames are queued here with NO synchronisation between them -- each to
its own result buffer, across changes of geometry, with a second net writing over the buffer the first one reads, through
the pipelined host route -- and every one must equal, byte for byte, the same frame run alone through the synchronous host
route.  (Round 5 tried the tail of frame k on a second stream beside the head of frame k + 1: 1.5 % SLOWER -- the two
kernels together move 560 MB in 200 us where one after the other they take 150, profiles/r05_ab_results.txt block 12 --
and was not kept; these tests are what held that experiment to the documented ordering, and hold whatever comes next.)"""
import numpy as np
import pytest

try:  # torch first: it must bring up its own HIP runtime before libuva.so pulls in /opt/rocm's
    import torch  # noqa: F401
except Exception:  # noqa: BLE001
    torch = None

from conftest import load_net

pytestmark = pytest.mark.gpu


@pytest.fixture(scope="module")
def nets(uva):
    assert uva.get_gpu_count() > 0, "no HIP device: the HIP path cannot run (no CPU fallback exists)"
    return {k: load_net(uva, k) for k in ("2x", "4x", "1x")}


def _dev(torch, a):
    return torch.from_numpy(np.ascontiguousarray(a)).cuda()


def _run_unsynchronised(torch, net, frames, tile):  # noqa: ARG001
    """every frame to its own result buffer, one synchronize at the end"""
    s = net.scale
    ins = [_dev(torch, f) for f in frames]
    outs = [torch.zeros((f.shape[0] * s, f.shape[1] * s, 3), dtype=torch.uint8, device="cuda") for f in frames]
    torch.cuda.synchronize()
    for f, i, o in zip(frames, ins, outs):
        net.process_u8_device(i.data_ptr(), f.shape[0], f.shape[1], o.data_ptr(), tile_size=tile, border=10)
    net.synchronize()
    return [o.cpu().numpy() for o in outs]


@pytest.mark.parametrize("key", ["2x", "4x"])
def test_frames_queued_back_to_back_equal_frames_run_alone(uva, nets, oracle, key):
    torch = pytest.importorskip("torch")
    plain = load_net(uva, key)
    net = nets[key]
    # 14 different frames of one geometry back to back
    frames = [oracle.synthetic_frame(270, 480, kind="random" if k & 1 else "smooth", seed=4000 + k) for k in range(14)]
    want = [plain.process_u8(f, tile_size=240, border=10) for f in frames]
    got = _run_unsynchronised(torch, net, frames, 240)
    for k, (g, w) in enumerate(zip(got, want)):
        assert np.array_equal(g, w), (key, k, float((g != w).mean()))
    # the geometry changes from frame to frame (another workspace each time)
    sizes = [(96, 128), (270, 480), (64, 64), (270, 480), (96, 128), (131, 77), (131, 77), (64, 64)]
    frames = [oracle.synthetic_frame(h, w, kind="random", seed=4100 + k) for k, (h, w) in enumerate(sizes)]
    want = [plain.process_u8(f, tile_size=64, border=10) for f in frames]
    got = _run_unsynchronised(torch, net, frames, 64)
    for k, (g, w) in enumerate(zip(got, want)):
        assert np.array_equal(g, w), (key, "geometry", k, float((g != w).mean()))


def test_1080p_stream_of_frames(uva, nets, oracle):
    """the headline geometry: 24 frames of 1080p, reference tiling, one result buffer per frame, one synchronize"""
    torch = pytest.importorskip("torch")
    plain = load_net(uva, "2x")
    base = oracle.synthetic_frame(1080, 1920, seed=31)
    frames = [np.roll(base, 37 * k, axis=1) ^ np.uint8(k) for k in range(24)]
    want = [plain.process_u8(f, tile_size=960, border=10) for f in frames]
    got = _run_unsynchronised(torch, nets["2x"], frames, 960)
    for k, (g, w) in enumerate(zip(got, want)):
        assert np.array_equal(g, w), (k, float((g != w).mean()))


def test_calls_behind_a_device_frame_see_the_whole_frame(uva, nets, oracle):
    """Each of the calls below is documented to come AFTER what the net has been asked so far: a second net through wait_for
    (it reads the first one's result), the host route on the same net, the debug tap."""
    torch = pytest.importorskip("torch")
    net, other = nets["2x"], load_net(uva, "2x")
    h, w = 540, 960
    a = oracle.synthetic_frame(h, w, seed=51)
    b = oracle.synthetic_frame(h, w, kind="random", seed=52)
    want_a = net.process_u8(a, tile_size=0)
    want_aa = net.process_u8(want_a[:h, :w].copy(), tile_size=0)
    want_b = net.process_u8(b, tile_size=0)
    for rep in range(6):
        d_in = _dev(torch, a)
        d_out = torch.zeros((2 * h, 2 * w, 3), dtype=torch.uint8, device="cuda")
        d_out2 = torch.zeros((2 * h, 2 * w, 3), dtype=torch.uint8, device="cuda")
        torch.cuda.synchronize()
        net.process_u8_device(d_in.data_ptr(), h, w, d_out.data_ptr(), tile_size=0)
        # `other` reads the top-left quarter of net's result (strided)
        other.wait_for(net)
        other.process_u8_device(d_out.data_ptr(), h, w, d_out2.data_ptr(), tile_size=0, in_stride=2 * w * 3)
        other.synchronize()
        assert np.array_equal(d_out2.cpu().numpy(), want_aa), rep
        net.synchronize()
        assert np.array_equal(d_out.cpu().numpy(), want_a), rep
        # the host route right behind a device frame: same net, same workspace, same activation buffers
        net.process_u8_device(d_in.data_ptr(), h, w, d_out.data_ptr(), tile_size=0)
        got_b = net.process_u8(b, tile_size=0)
        assert np.array_equal(got_b, want_b), rep
        net.synchronize()
        assert np.array_equal(d_out.cpu().numpy(), want_a), rep
        # the debug tap replays the last call
        net.process_u8_device(d_in.data_ptr(), h, w, d_out.data_ptr(), tile_size=0)
        act = net.debug_read_activation(3, h, w)
        assert np.isfinite(act).all()
        net.synchronize()
        assert np.array_equal(d_out.cpu().numpy(), want_a), rep


def test_a_second_net_may_overwrite_the_input_after_wait_for(uva, nets, oracle):
    """config 3's shape: `pre` writes the frame the 2x net reads; pre.wait_for(net) is what allows it to write the next one
    into the same buffer -- the 2x net's tail, its last kernel, still reads that buffer (the residual)."""
    torch = pytest.importorskip("torch")
    net, pre = nets["2x"], nets["1x"]
    h, w = 360, 640
    frames = [oracle.synthetic_frame(h, w, kind="random" if k & 1 else "smooth", seed=70 + k) for k in range(10)]
    want = [net.process_u8(pre.process_u8(f, tile_size=0), tile_size=320, border=10) for f in frames]
    ins = [_dev(torch, f) for f in frames]
    mid = torch.zeros((h, w, 3), dtype=torch.uint8, device="cuda")          # ONE buffer between the two nets
    outs = [torch.zeros((2 * h, 2 * w, 3), dtype=torch.uint8, device="cuda") for _ in frames]
    torch.cuda.synchronize()
    for i, o in zip(ins, outs):
        pre.wait_for(net)
        pre.process_u8_device(i.data_ptr(), h, w, mid.data_ptr(), tile_size=0)
        net.wait_for(pre)
        net.process_u8_device(mid.data_ptr(), h, w, o.data_ptr(), tile_size=320, border=10)
    net.synchronize()
    for k, (o, wnt) in enumerate(zip(outs, want)):
        assert np.array_equal(o.cpu().numpy(), wnt), k


def test_pipelined_host_route_three_frames_in_flight(uva, nets, oracle):
    """submit / collect: the download of frame k waits for its last kernel, the upload of frame k + 2 for nothing"""
    plain = load_net(uva, "2x")
    net = nets["2x"]
    frames = [oracle.synthetic_frame(270, 480, kind="random", seed=90 + k) for k in range(9)]
    want = [plain.process_u8(f, tile_size=240, border=10) for f in frames]
    tickets, got = [], []
    for f in frames:
        if len(tickets) == 3:
            got.append(net.collect_u8(tickets.pop(0)))
        tickets.append(net.submit_u8(f, tile_size=240, border=10))
    while tickets:
        got.append(net.collect_u8(tickets.pop(0)))
    for k, (g, w) in enumerate(zip(got, want)):
        assert np.array_equal(g, w), k


def test_denoise_between_device_frames(uva, nets, oracle):
    """`-m n=K` between two device frames: the stage overwrites the buffer the previous frame's tail reads"""
    torch = pytest.importorskip("torch")
    net = nets["2x"]
    h, w = 180, 320
    frames = [oracle.synthetic_frame(h, w, kind="random", seed=120 + k) for k in range(6)]
    buf = torch.zeros((h, w, 3), dtype=torch.uint8, device="cuda")          # the denoised frame: one buffer for all frames
    ins = [_dev(torch, f) for f in frames]
    outs = [torch.zeros((2 * h, 2 * w, 3), dtype=torch.uint8, device="cuda") for _ in frames]
    torch.cuda.synchronize()
    want = []
    for i in ins:                   # frame by frame, synchronised
        net.denoise_u8_device(i.data_ptr(), h, w, buf.data_ptr(), 3.0, after=net)
        o = torch.zeros((2 * h, 2 * w, 3), dtype=torch.uint8, device="cuda")
        net.process_u8_device(buf.data_ptr(), h, w, o.data_ptr(), tile_size=0)
        net.synchronize()
        want.append(o.cpu().numpy())
    for i, o in zip(ins, outs):     # ... and back to back
        net.denoise_u8_device(i.data_ptr(), h, w, buf.data_ptr(), 3.0, after=net)
        net.process_u8_device(buf.data_ptr(), h, w, o.data_ptr(), tile_size=0)
    net.synchronize()
    for k, (o, wnt) in enumerate(zip(outs, want)):
        assert np.array_equal(o.cpu().numpy(), wnt), k
