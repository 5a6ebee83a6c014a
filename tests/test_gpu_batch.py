"""-m gpu: uva_net_process_u8_device_batch (round 6): several frames of one geometry per call.  The 1x HurrDeblur net takes up to
eight of them per sub10_kernel launch -- the strips' warm-up rows and the pipeline's fill and drain are paid once per launch --,
every other net runs them frame by frame.  What is asked for: EXACTLY the bytes of one uva_net_process_u8_device call per frame
(which the other tests pin to the oracle), for every count, for sizes whose batches do not fit the kernel's row table, for
padded strides, and for frames that differ from each other (a frame index that went to the wrong buffer would show)."""
import numpy as np
import pytest

try:
    import torch
except Exception:  # noqa: BLE001
    torch = None

from conftest import load_net
from parity_report import check_u8, fp32_bar

pytestmark = pytest.mark.gpu


def run_batch(net, frames, tile_size=0, border=0):
    """frames -> results through ONE batch call (device buffers made here)"""
    h, w = frames[0].shape[:2]
    s = net.scale
    d_in = [torch.from_numpy(f).cuda() for f in frames]
    d_out = [torch.full((h * s, w * s, 3), 7, dtype=torch.uint8, device="cuda") for _ in frames]
    torch.cuda.synchronize()
    net.process_u8_device_batch([t.data_ptr() for t in d_in], h, w, [t.data_ptr() for t in d_out], tile_size=tile_size, border=border)
    net.synchronize()
    return [t.cpu().numpy() for t in d_out]


@pytest.fixture(scope="module")
def net1x(uva):
    assert uva.get_gpu_count() > 0
    return load_net(uva, "1x")


@pytest.mark.parametrize("h,w", [(50, 33), (1, 1), (37, 121), (300, 700), (1080, 1920)])
@pytest.mark.parametrize("count", [1, 2, 3, 5, 8, 11])
def test_a_batch_gives_the_bytes_of_the_single_frame_calls(net1x, oracle, h, w, count):
    if h * w > 10 ** 6 and count not in (2, 4, 5):
        pytest.skip("full-size frames: the counts that matter (2, 4 + 1)")
    frames = [oracle.synthetic_frame(h, w, kind="random" if k & 1 else "smooth", seed=77 * count + k) for k in range(count)]
    want = [net1x.process_u8(f, tile_size=0) for f in frames]
    got = run_batch(net1x, frames)
    for k in range(count):
        assert np.array_equal(got[k], want[k]), (h, w, count, k, float((got[k] != want[k]).mean()))


def test_full_size_batch_of_four_against_the_oracle_in_windows(net1x, oracle_models, oracle):
    h, w, rad, win = 1080, 1920, 10, 24
    frames = [oracle.synthetic_frame(h, w, seed=500 + k) for k in range(4)]
    got = run_batch(net1x, frames)
    for k, (y0, x0) in enumerate([(0, 0), (h - win, w - win), (h // 2, 930), (270, 60)]):      # (one window per frame: segment and strip edges)
        cy0, cx0, cy1, cx1 = max(0, y0 - rad), max(0, x0 - rad), min(h, y0 + win + rad), min(w, x0 + win + rad)
        want = oracle_models["1x"].apply_model(np.ascontiguousarray(frames[k][cy0:cy1, cx0:cx1]))[y0 - cy0:y0 - cy0 + win, x0 - cx0:x0 - cx0 + win]
        check_u8(f"1x batch of 4, frame {k}, 1080p window ({y0},{x0})", np.ascontiguousarray(got[k][y0:y0 + win, x0:x0 + win]),
                 np.ascontiguousarray(want), vs="fp32 oracle", max_lsb=1, min_psnr=55, model="1x", route="whole")


def test_padded_strides_and_aliasing_free_outputs(net1x, oracle):
    h, w, count = 41, 130, 3
    frames = [oracle.synthetic_frame(h, w, seed=9 + k) for k in range(count)]
    want = [net1x.process_u8(f, tile_size=0) for f in frames]
    ins, outs = [], []
    for f in frames:
        t = torch.zeros((h, w * 3 + 37), dtype=torch.uint8, device="cuda")
        t[:, :w * 3] = torch.from_numpy(f.reshape(h, w * 3)).cuda()
        ins.append(t)
        outs.append(torch.full((h, w * 3 + 91), 201, dtype=torch.uint8, device="cuda"))
    torch.cuda.synchronize()
    net1x.process_u8_device_batch([t.data_ptr() for t in ins], h, w, [t.data_ptr() for t in outs], in_stride=w * 3 + 37, out_stride=w * 3 + 91)
    net1x.synchronize()
    for k in range(count):
        o = outs[k].cpu().numpy()
        assert np.array_equal(o[:, :w * 3].reshape(h, w, 3), want[k])
        assert (o[:, w * 3:] == 201).all()                     # nothing written past a row's end


def test_a_frame_too_large_for_a_batch_of_eight_splits_itself(net1x, oracle):
    """2160p: one frame per launch fits sub10_kernel's row table, two do not -- the call must find that out and still return
    every frame's bytes"""
    h, w = 2160, 3840
    frames = [oracle.synthetic_frame(h, w, seed=3 + k) for k in range(2)]
    want = [net1x.process_u8(f, tile_size=0) for f in frames]
    got = run_batch(net1x, frames)
    assert all(np.array_equal(g, x) for g, x in zip(got, want))


def test_other_nets_run_frame_by_frame(uva, oracle):
    net = load_net(uva, "2x")
    frames = [oracle.synthetic_frame(70, 75, seed=40 + k) for k in range(3)]
    want = [net.process_u8(f, tile_size=32, border=10) for f in frames]
    got = run_batch(net, frames, tile_size=32, border=10)
    assert all(np.array_equal(g, x) for g, x in zip(got, want))


def test_bad_arguments_are_refused(net1x):
    from upscale_video_amd._lib import UvaError
    with pytest.raises(UvaError):
        net1x.process_u8_device_batch([0], 8, 8, [0])
    net1x.process_u8_device_batch([], 8, 8, [])                  # nothing to do is not an error
