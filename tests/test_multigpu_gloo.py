"""N > 1 path of bench.py on CPU: two gloo ranks shard frames as independent units (no data-path
collective), and the timing protocol (barrier + sync both sides, MAX over ranks) aggregates to a
whole-job rate.  The per-frame work here is the CPU oracle on tiny frames (test infrastructure);
on the GPU box the same protocol wraps uva_net_process_u8_device."""
import os
import sys

import numpy as np
import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _worker(rank, world, port, n_frames, q):
    sys.path.insert(0, ROOT)
    import time
    import torch
    import torch.distributed as dist
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port), RANK=str(rank), WORLD_SIZE=str(world))
    dist.init_process_group("gloo", rank=rank, world_size=world)
    import bench
    from oracle import uvoracle
    m = uvoracle.load_model("1x")
    mine = bench.shard_frames(n_frames, rank, world)
    outs = {}

    def run():
        for f in mine:
            outs[f] = m.apply_model(uvoracle.synthetic_frame(16, 24, seed=f), threads=1)
        if rank == 1:
            time.sleep(0.2)          # the slow rank must set the job time

    def max_over_ranks(x):
        t = torch.tensor([x], dtype=torch.float64)
        dist.all_reduce(t, op=dist.ReduceOp.MAX)
        return float(t.item())

    elapsed = bench.timed_region(run, lambda: None, dist.barrier, max_over_ranks)
    # second region, as bench.py runs the PCIe-inclusive host route on every rank: the same protocol, its
    # own barriers, the fast rank of the first region is the slow one here
    host = bench.timed_region(lambda: time.sleep(0.3 if rank == 0 else 0.05), lambda: None, dist.barrier, max_over_ranks)
    assert host >= 0.3 and abs(bench.whole_job_rate(10, world, host) - 10 * world / host) < 1e-9
    checksum = float(sum(int(v.astype(np.int64).sum()) for v in outs.values()))
    t = torch.tensor([checksum, float(len(mine))], dtype=torch.float64)
    dist.all_reduce(t)               # test bookkeeping only, not part of the data path
    q.put((rank, mine, elapsed, float(t[0]), int(t[1])))
    dist.destroy_process_group()


def test_frame_sharding_two_ranks_gloo():
    import torch.multiprocessing as mp
    sys.path.insert(0, ROOT)
    from oracle import uvoracle
    uvoracle.build()
    n_frames, world, port = 7, 2, 29517 + os.getpid() % 1000
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    procs = [ctx.Process(target=_worker, args=(r, world, port, n_frames, q)) for r in range(world)]
    for p in procs:
        p.start()
    res = sorted(q.get(timeout=120) for _ in procs)
    for p in procs:
        p.join(timeout=60)
        assert p.exitcode == 0
    (r0, f0, e0, c0, n0), (r1, f1, e1, c1, n1) = res
    assert f0 == [0, 2, 4, 6] and f1 == [1, 3, 5]           # disjoint, complete, no exchange
    assert n0 == n1 == n_frames
    assert e0 == e1 and e0 >= 0.2                          # MAX over ranks: the slow rank decides
    m = uvoracle.load_model("1x")
    want = float(sum(int(m.apply_model(uvoracle.synthetic_frame(16, 24, seed=f), threads=1).astype(np.int64).sum())
                     for f in range(n_frames)))
    assert c0 == c1 == want                                # every frame processed exactly once


def test_shard_frames_properties():
    sys.path.insert(0, ROOT)
    import bench
    for n in (0, 1, 5, 64):
        for world in (1, 2, 4, 8):
            parts = [bench.shard_frames(n, r, world) for r in range(world)]
            flat = sorted(f for p in parts for f in p)
            assert flat == list(range(n))
            assert max(len(p) for p in parts) - min(len(p) for p in parts) <= 1


def test_conv_flops_match_baseline_table():
    sys.path.insert(0, ROOT)
    import bench
    assert bench.conv_flops_per_px(64, 18, 2) == 1196928
    assert bench.conv_flops_per_px(64, 18, 4) == 1238400
    assert bench.conv_flops_per_px(24, 10, 1) == 85536


def _fork_worker(rank, world, barrier, slots, q):
    """a self-launched rank of `python bench.py --gpus N`: bench.ForkComm, no torch.distributed at all"""
    sys.path.insert(0, ROOT)
    import time
    import bench
    comm = bench.ForkComm(rank, world, barrier, slots)
    assert "torch.distributed" not in sys.modules
    # the slow rank sets the job time, in both regions, and every rank gets the same number
    a = bench.timed_region(lambda: time.sleep(0.25 if rank == 1 else 0.02), lambda: None, comm.barrier, comm.max_over_ranks)
    b = bench.timed_region(lambda: time.sleep(0.3 if rank == 0 else 0.02), lambda: None, comm.barrier, comm.max_over_ranks)
    q.put((rank, a, b, bench.shard_frames(7, rank, world)))


def test_self_launched_ranks_fence_without_rccl():
    """python bench.py --gpus N starts its own ranks: barrier + MAX over ranks through multiprocessing primitives"""
    import multiprocessing as mp
    world = 3
    ctx = mp.get_context("spawn")
    barrier, slots, q = ctx.Barrier(world), ctx.Array("d", world), ctx.Queue()
    procs = [ctx.Process(target=_fork_worker, args=(r, world, barrier, slots, q)) for r in range(world)]
    for p in procs:
        p.start()
    res = sorted(q.get(timeout=120) for _ in range(world))
    for p in procs:
        p.join(60)
        assert p.exitcode == 0
    assert len({round(r[1], 9) for r in res}) == 1 and len({round(r[2], 9) for r in res}) == 1      # one number for the job
    assert 0.25 <= res[0][1] < 0.6 and 0.3 <= res[0][2] < 0.7
    assert sorted(f for r in res for f in r[3]) == list(range(7))


def _dying_rank(rank, world, device, barrier, slots, argv):
    """a rank of launch_ranks() that gets through one timed region and then loses rank 1 in the middle of the second --
    what an out-of-memory kill, a missing device or a HIP error does to `python bench.py --gpus N`"""
    sys.path.insert(0, ROOT)
    import time
    import bench
    comm = bench.ForkComm(rank, world, barrier, slots)
    bench.timed_region(lambda: time.sleep(0.05), lambda: None, comm.barrier, comm.max_over_ranks)
    how = argv[1]

    def region():
        if rank == 1:
            time.sleep(0.2)
            if how == "kill":
                os.kill(os.getpid(), 9)
            if how == "exit":
                os._exit(7)
            raise RuntimeError("rank 1 fails")
        time.sleep(0.05)
    bench.timed_region(region, lambda: None, comm.barrier, comm.max_over_ranks)       # ranks 0 and 2 wait at the fence: RankLost
    raise AssertionError("rank %d got through a fence that rank 1 never reached" % rank)


@pytest.mark.parametrize("how", ["kill", "exit", "raise"])
def test_a_dead_rank_ends_the_job_within_seconds(how, capfd):
    """a rank that dies before a fence must not leave the others in wait() for ever (the driver's 8-GPU
    command would burn its whole timeout and report nothing): the parent sees the exit code, aborts the barrier, stops the
    rest and returns non-zero -- here in well under 30 s, with the fences' own timeout (900 s) never reached."""
    import time
    sys.path.insert(0, ROOT)
    import bench
    t0 = time.monotonic()
    rc = bench.launch_ranks(3, [0, 1, 2], ["bench.py", how], target=_dying_rank)
    dt = time.monotonic() - t0
    assert rc != 0 and dt < 30, (rc, dt)
    assert rc == {"kill": -9, "exit": 7, "raise": 1}[how]
    err = capfd.readouterr().err
    assert "rank 1" in err and "fence was aborted" in err
    assert "got through a fence" not in err            # nobody passed the broken fence
    assert "RankLost" in err                           # the survivors left their wait with the named error


def _ok_rank(rank, world, device, barrier, slots, argv):
    sys.path.insert(0, ROOT)
    import time
    import bench
    comm = bench.ForkComm(rank, world, barrier, slots)
    for _ in range(3):
        bench.timed_region(lambda: time.sleep(0.01 * (rank + 1)), lambda: None, comm.barrier, comm.max_over_ranks)


def test_eight_healthy_ranks_return_zero():
    """the documented N = 8 shape (python bench.py --gpus 8) through the same launcher: eight ranks, three fenced regions, rc 0"""
    sys.path.insert(0, ROOT)
    import bench
    assert bench.launch_ranks(8, list(range(8)), ["bench.py"], target=_ok_rank) == 0


def test_a_fence_times_out_on_its_own():
    """the backstop under the parent's watch: a Barrier wait with a timeout breaks the barrier for everybody"""
    import multiprocessing as mp
    sys.path.insert(0, ROOT)
    import bench
    ctx = mp.get_context("spawn")
    comm = bench.ForkComm(0, 2, ctx.Barrier(2), ctx.Array("d", 2), timeout=0.3)
    with pytest.raises(bench.RankLost):
        comm.barrier()                                  # rank 1 never comes
    with pytest.raises(bench.RankLost):
        comm.max_over_ranks(1.0)                        # and the fence stays broken


def test_pinned_budget_check_says_no_before_anything_is_allocated(monkeypatch):
    sys.path.insert(0, ROOT)
    import bench
    need, avail = bench.pinned_budget_check(8, 8, 2160, 3840, 2)            # config 5: eight ranks, 3 frames in flight each
    assert need == 8 * 3 * (2160 * 3840 * 3 + 4320 * 7680 * 3)
    assert avail is None or avail > 0
    with pytest.raises(SystemExit, match="page-locked"):
        bench.pinned_budget_check(8, 8, 216000, 384000, 4)                  # 29 TB: no node has it


def test_bench_refuses_a_device_list_of_the_wrong_length():
    import subprocess
    r = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--gpus", "2", "--devices", "0"], capture_output=True, text=True,
                       env=dict(os.environ, UVA_LIB_PATH=os.path.join(ROOT, "upscale_video_amd", "libuva.so")))
    assert r.returncode != 0 and "--devices" in (r.stderr + r.stdout)


@pytest.mark.parametrize("tile", [64, 0])
def test_bench_parity_windows_are_exact_for_an_exact_result(tile):
    """bench.py's full-size parity (windows of the timed frame against the oracle on their receptive fields, cut to the window's
    tile where the reference tiles): with the oracle's own frame standing in for the GPU's the distance must be exactly zero --
    corners, centre and both sides of the seams, tiled and whole -- and a planted error must be found"""
    sys.path.insert(0, ROOT)
    import bench
    from oracle import uvoracle
    uvoracle.build()
    img = uvoracle.synthetic_frame(150, 200, seed=3)
    m = uvoracle.load_model("2x")
    got = m.upscale_image(img, tile_size=tile, border=10) if tile else m.apply_model(img)
    r = bench.parity_windows("2x", img, got, 2, 18, tile, win=24)
    assert r["max_abs_lsb"] == 0 and r["differ_share"] == 0.0 and r["windows"] == (9 if tile else 5)
    bad = got.copy()
    bad[0, 0, 0] ^= 4
    assert bench.parity_windows("2x", img, bad, 2, 18, tile, win=24)["max_abs_lsb"] == 4
