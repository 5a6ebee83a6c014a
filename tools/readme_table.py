"""The README's measurement table from a committed evidence set: python tools/readme_table.py profiles r06_final
(every cell names the file it comes from; nothing is typed by hand)."""
import json
import os
import sys

d, tag = sys.argv[1], sys.argv[2]


def line(name):
    p = os.path.join(d, f"{tag}_{name}.json")
    try:
        return json.loads(open(p).read().strip().splitlines()[-1]), os.path.basename(p)
    except Exception:  # noqa: BLE001
        return None, os.path.basename(p)


rows = []


def add(what, name, fmt):
    j, f = line(name)
    if j:
        rows.append((what, fmt(j), f))


K = lambda j: f"**{j['value']:.1f}**"                                     # noqa: E731
KR = lambda j: f"**{j['value']:.1f}** frames/s, `roofline.frac` {j['roofline']['frac']:.3f} ({j['roofline']['avg_launch_ms'] * 1e3:.1f} µs per launch)"   # noqa: E731
add("1080p → 2× Compact, reference 960/10 tiling, frames resident in HBM (`bench.py`'s `value`; the dominant kernel `trunkw_kernel<64>`)", "bench", KR)
add("… the driver's form (`--steps 20 --warmup 5`)", "bench_driver_form", KR)
add("… the whole frame (head, 8 trunk launches, tail) against the nominal fp16 MFMA peak", "bench",
    lambda j: f"{j['config']['whole_path_tflops']:.0f} TFLOP/s = {j['config']['whole_path_tflops'] / 2500:.3f}")
add("… host to host, page-locked frames, H2D / kernels / D2H pipelined (route E)", "bench", lambda j: f"{j['config']['host_route_fps_pcie_inclusive']:.1f} frames/s")
add("… one synchronous call per frame on pageable numpy arrays", "bench", lambda j: f"{j['config']['host_route_sync_pageable_fps']:.1f} frames/s")
add("… fp32 CPU oracle on the box's 16 granted cores (`cpu_baseline`, kind `port`)", "bench", lambda j: f"{j['cpu_baseline']['value']:.3f} frames/s")
add("1080p → 4× Compact", "bench_4x_compact_1080p", KR)
add("3840×2160 → 2× Compact (config 5's frame)", "bench_2x_compact_2160p", KR)
add("1× HurrDeblur, whole frame, four frames per launch (`--batch 4`, the workload's default)", "bench_1x_batch4", KR)
add("1× HurrDeblur, one frame per call", "bench_1x_batch1", KR)
add("chain 1× → u8 → 2× (config 3)", "bench_chain_1x_2x_1080p", lambda j: f"**{j['value']:.1f}** frames/s")
add("`4x_Valar_v1` 1080p → 8K, synthetic weights (config 4 as named; `rdb4_kernel`)", "bench_4x_valar_1080p", KR)
add("two ranks on the one GPU, dynamic frame queue", "bench_two_ranks_one_gpu_dynamic", lambda j: f"{j['value']:.1f} frames/s ({' + '.join(str(r['frames_K']) for r in j['per_rank'])} frames)")
add("config 5's line, eight ranks on the one GPU at 3840×2160", "bench_config5_eight_ranks_one_gpu", lambda j: f"{j['value']:.1f} frames/s")
print("| what | measured | file (`profiles/`) |\n|---|---|---|")
for what, val, f in rows:
    print(f"| {what} | {val} | `{f}` |")
