#!/usr/bin/env python3
"""Two builds of the library must give the same bytes: every frame of a set (1080p with the reference tiling through the 2x and 4x
nets, whole frames, small tiles, odd sizes) through each build in its own process, SHA-256 of every result compared.
    python tools/lib_identity.py upscale_video_amd/libuva.so upscale_video_amd/libuva_<variant>.so"""
import hashlib
import json
import os
import subprocess
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
CASES = [("2x_Compact_Pretrain", 1080, 1920, 960), ("4x_Compact_Pretrain", 1080, 1920, 960), ("2x_Compact_Pretrain", 1080, 1920, 0),
         ("2x_Compact_Pretrain", 2160, 3840, 960), ("2x_Compact_Pretrain", 200, 190, 64), ("4x_Compact_Pretrain", 131, 77, 32),
         ("2x_Compact_Pretrain", 110, 122, 55), ("2x_Compact_Pretrain", 64, 64, 0), ("2x_Compact_Pretrain", 33, 1000, 0),
         ("4x_Compact_Pretrain", 720, 1280, 960), ("2x_Compact_Pretrain", 1000, 9, 0), ("2x_Compact_Pretrain", 540, 960, 240)]


HURR = "1x_HurrDeblur_SubCompact_nf24-nc8_244k_net_g"
CASES_1X = [(HURR, 1080, 1920, 0), (HURR, 720, 1280, 0), (HURR, 300, 700, 0), (HURR, 61, 59, 0), (HURR, 1, 1, 0), (HURR, 9, 1000, 0),
            (HURR, 1000, 9, 0), (HURR, 137, 241, 0), (HURR, 2160, 3840, 0), (HURR, 64, 60, 0), (HURR, 23, 121, 0), (HURR, 500, 180, 0)]


def cases():
    """the fixed set, plus UVA_IDENTITY_RANDOM seeded random geometries (frame size, tile size: few-step and many-step workgroups);
    UVA_IDENTITY_1X=1: whole frames through the 1x net instead (sub10_kernel), UVA_IDENTITY_RANDOM more of them"""
    if os.environ.get("UVA_IDENTITY_1X"):
        out = list(CASES_1X)
        n = int(os.environ.get("UVA_IDENTITY_RANDOM", "0"))
        if n:
            import numpy as np
            rng = np.random.default_rng(2027)
            out += [(HURR, int(rng.integers(1, 1200)), int(rng.integers(1, 2000)), 0) for _ in range(n)]
        return out
    out = list(CASES)
    n = int(os.environ.get("UVA_IDENTITY_RANDOM", "0"))
    if n:
        import numpy as np
        rng = np.random.default_rng(2026)
        for _ in range(n):
            stem = "4x_Compact_Pretrain" if rng.integers(0, 4) == 0 else "2x_Compact_Pretrain"
            h, w = int(rng.integers(1, 700)), int(rng.integers(1, 900))
            tile = int(rng.choice([0, 0, 32, 48, 64, 100, 240, 960]))
            if tile and -(-h // tile) * -(-w // tile) > 64:      # (the library's limit: 64 tiles per frame)
                tile = 240
            out.append((stem, h, w, tile))
    return out


def child():
    sys.path.insert(0, ROOT)
    import numpy as np  # noqa: F401
    from upscale_video_amd import ncnn
    from upscale_video_amd.synth import synthetic_frame
    out = []
    nets = {}
    for rep in range(int(os.environ.get("UVA_IDENTITY_REPS", "2"))):
        for k, (stem, h, w, tile) in enumerate(cases()):
            if stem not in nets:
                n = ncnn.Net()
                n.set_vulkan_device(0)
                base = os.path.join(ROOT, "models", stem)
                assert n.load_param(base + ".param") == 0 and n.load_model(base + ".bin") == 0
                nets[stem] = n
            img = synthetic_frame(h, w, seed=100 * rep + k, kind="random" if (k + rep) & 1 else "smooth")
            res = nets[stem].process_u8(img, tile_size=tile, border=10 if tile else 0)
            out.append(hashlib.sha256(res.tobytes()).hexdigest())
    print(json.dumps(out))


if __name__ == "__main__":
    if len(sys.argv) == 2 and sys.argv[1] == "--child":
        child()
        sys.exit(0)
    libs = [os.path.abspath(p) for p in sys.argv[1:]]
    res = []
    for lib in libs:
        env = dict(os.environ, UVA_LIB_PATH=lib)
        r = subprocess.run([sys.executable, os.path.abspath(__file__), "--child"], env=env, capture_output=True, text=True, timeout=1200)
        if r.returncode:
            print(r.stderr[-2000:])
            sys.exit(1)
        res.append(json.loads(r.stdout.strip().splitlines()[-1]))
    bad = 0
    for lib, r in zip(libs[1:], res[1:]):
        diff = [i for i, (a, b) in enumerate(zip(res[0], r)) if a != b]
        print(f"{os.path.basename(lib)} vs {os.path.basename(libs[0])}: {len(r)} frames, {len(diff)} differ {diff[:10]}")
        bad += len(diff)
    sys.exit(1 if bad else 0)
