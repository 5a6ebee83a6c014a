"""Package power, clock and time per launch of the trunk kernel and its ablations in the sustained
(power-limited) state: each variant is launched back to back for several seconds while rocm-smi is
sampled from a second thread.  energy/launch = mean power x time/launch.  GPU only, debug aid."""
import ctypes, os, re, subprocess, sys, threading, time
import numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from upscale_video_amd import _lib, ncnn  # noqa: E402

net = ncnn.Net(); net.set_vulkan_device(0)
base = os.path.join(ROOT, "models", "2x_Compact_Pretrain")
assert net.load_param(base + ".param") == 0 and net.load_model(base + ".bin") == 0
from upscale_video_amd.synth import synthetic_frame  # noqa: E402
kind = sys.argv[1] if len(sys.argv) > 1 else "smooth"
img = synthetic_frame(1080, 1920) if kind == "smooth" else np.zeros((1080, 1920, 3), np.uint8)
net.process_u8(img, tile_size=960, border=10)


def smi():
    out = subprocess.run(["rocm-smi", "--showpower", "--showclocks"], capture_output=True, text=True).stdout
    p = re.search(r"Power \(W\): ([\d.]+)", out)
    c = re.search(r"sclk clock level: \S+ \((\d+)Mhz\)", out)
    return (float(p.group(1)) if p else float("nan"), float(c.group(1)) if c else float("nan"))


print(f"frame: {kind}; idle: {smi()}")
L = _lib.load()
for ablate, label in ((0, "real kernel"), (2, "compute only (L2-resident in, sink out)"), (4, "memory + LDS fragment reads, no MFMA"),
                      (1, "memory only (no MFMA, no LDS reads)")):
    samples, stop = [], threading.Event()

    def sampler():
        time.sleep(1.5)
        while not stop.is_set():
            samples.append(smi())
            time.sleep(0.3)

    th = threading.Thread(target=sampler); th.start()
    res = []
    t0 = time.time()
    while time.time() - t0 < 6.0:
        buf = np.zeros(256 * 8, np.uint64); n = ctypes.c_int(); ms = ctypes.c_float()
        _lib.check(L.uva_net_debug_trunk_stamps(net._h, buf.ctypes.data, 256, n, ablate | (20 << 8), ms))
        res.append(ms.value)
    stop.set(); th.join()
    us = float(np.median(res[1:])) * 1e3
    pw = float(np.nanmean([s[0] for s in samples])); ck = float(np.nanmean([s[1] for s in samples]))
    print(f"{label:44s} {us:7.1f} us/launch  {pw:6.0f} W  sclk {ck:5.0f} MHz  => {pw * us * 1e-3:6.1f} mJ/launch  ({len(samples)} samples)")
