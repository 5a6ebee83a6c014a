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


# ---- round 6: per-rank records, the shared frame counter (--dynamic) ----------------------------------------------------------
def _record_rank(rank, world, device, barrier, slots, argv):
    """a rank of launch_ranks(): pulls frames off the ONE counter (a slow rank takes fewer), then the ranks exchange records"""
    sys.path.insert(0, ROOT)
    import time
    import bench
    comm = bench.ForkComm(rank, world, barrier, slots)
    cpath = os.path.join(bench.job_scratch_dir(), "frame_counter")
    if rank == 0:
        bench.FileCounter(cpath, create=True).close()
    comm.barrier()
    counter = bench.FileCounter(cpath)
    total, mine = 64, []

    def region():
        while True:
            i = counter.take(2)
            if i >= total:
                break
            mine.extend(range(i, min(i + 2, total)))
            time.sleep(0.02 if rank == 1 else 0.002)        # rank 1 is the slow GPU
    own = []
    elapsed = bench.timed_region(region, lambda: None, comm.barrier, comm.max_over_ranks, own=own)
    assert 0 < own[0] <= elapsed + 1e-6
    rec = {"rank": rank, "device": device, "numa_node": rank % 2, "frames_K": len(mine), "fps_K": len(mine) / own[0], "fps_E": None,
           "h2d_GBps": 50.0 + rank, "d2h_GBps": 40.0, "first_frame_ms": 12.5, "frames": mine}
    records = bench.gather_records(comm, rec)
    assert [r["rank"] for r in records] == list(range(world))
    flat = sorted(f for r in records for f in r["frames"])
    assert flat == list(range(total)), "every frame taken exactly once"
    assert records[1]["frames_K"] < min(r["frames_K"] for r in records if r["rank"] != 1), "the slow rank took fewer frames"
    summary = bench.per_rank_summary(records)
    assert summary["h2d_GBps"] == {"min": 50.0, "max": 50.0 + world - 1, "mean": round(50.0 + (world - 1) / 2, 3)}
    assert "rank" not in summary and "numa_node" not in summary and summary["frames_K"]["max"] > summary["frames_K"]["min"]
    if rank == 0:
        with open(argv[1], "w") as f:
            import json
            json.dump({"per_rank": records, "per_rank_summary": summary}, f)


def test_eight_ranks_share_one_frame_counter_and_report_per_rank(tmp_path, monkeypatch):
    """the first real 8-GPU run must explain itself -- per-rank records (device, NUMA node, K and E rates,
    PCIe rates, first-frame time) in the one JSON line, min / max / mean over ranks, and a --dynamic mode in which the ranks pull
    frame indices from one shared counter.  Eight self-launched ranks on CPU: schema, exactly-once, and the slow rank takes less."""
    import json
    sys.path.insert(0, ROOT)
    import bench
    monkeypatch.delenv("UVA_BENCH_JOB_DIR", raising=False)
    out = tmp_path / "line.json"
    assert bench.launch_ranks(8, list(range(8)), ["bench.py", str(out)], target=_record_rank) == 0
    line = json.loads(out.read_text())
    assert len(line["per_rank"]) == 8
    for r in line["per_rank"]:
        assert {"rank", "device", "numa_node", "frames_K", "fps_K", "fps_E", "h2d_GBps", "d2h_GBps", "first_frame_ms"} <= set(r)
    assert sum(r["frames_K"] for r in line["per_rank"]) == 64
    assert set(line["per_rank_summary"]) >= {"fps_K", "h2d_GBps", "d2h_GBps", "first_frame_ms", "frames_K"}


def test_file_counter_is_exact_under_contention(tmp_path):
    import threading
    sys.path.insert(0, ROOT)
    import bench
    path = str(tmp_path / "c")
    bench.FileCounter(path, create=True).close()
    got = []

    def worker():
        c = bench.FileCounter(path)
        for _ in range(200):
            got.append(c.take(3))
        c.close()
    ts = [threading.Thread(target=worker) for _ in range(6)]
    [t.start() for t in ts]
    [t.join() for t in ts]
    assert sorted(got) == list(range(0, 3 * 1200, 3))
    c = bench.FileCounter(path)
    c.reset()
    assert c.take() == 0 and c.take() == 1


def test_harness_prints_per_worker_rates():
    """test_gpus.py -g 0,..,7: the calls' seconds grouped per worker (slot, GPU), slowest and fastest named"""
    sys.path.insert(0, ROOT)
    import test_gpus
    times = {(k, str(k % 4)): [0.1 + 0.01 * k] * (3 if k else 2) for k in range(8)}
    lines = test_gpus.per_worker_table(times, elapsed=0.5)
    assert len(lines) == 1 + 8 + 1
    assert "worker 0 on GPU 0: 2 calls, 0.1000 s per call, 10.00 frames/s" in lines[1]
    assert "worker 7 on GPU 3: 3 calls" in lines[8]
    assert "slowest / fastest worker: 5.88 / 10.00 frames/s" in lines[-1] and "the pool's 46.00" in lines[-1]
