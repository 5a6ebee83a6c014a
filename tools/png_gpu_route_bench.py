"""The PNG route with the result frame deflated on the GPU (csrc/uva_png.hip.h) against the same route with zlib on the
host's cores: file-to-file frames/s of the persistent FramePool workers (1080p '<n>.extract.png' -> 2x -> 3840x2160
'<n>.png' in a RAM disk), plus the encoder by itself.  usage: python tools/png_gpu_route_bench.py [frames=240]"""
import os
import subprocess
import sys
import time

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)


def encoder_alone():
    import numpy as np
    from upscale_video_amd import _imageio, ncnn
    from upscale_video_amd.synth import synthetic_frame
    small = synthetic_frame(1080, 1920, seed=1)
    net = ncnn.Net()
    net.set_vulkan_device(0)
    base = os.path.join(ROOT, "models", "2x_Compact_Pretrain")
    assert net.load_param(base + ".param") == 0 and net.load_model(base + ".bin") == 0
    big = net.process_u8(small, tile_size=960, border=10)                 # a real result frame: what imwrite sees
    ws = ncnn.PngWorkspace(2160, 3840)
    ncnn.png_encode_u8(big, workspace=ws)
    t0 = time.perf_counter()
    for _ in range(10):
        png = ncnn.png_encode_u8(big, workspace=ws)
    t_gpu = (time.perf_counter() - t0) / 10
    t0 = time.perf_counter()
    for _ in range(10):
        ws.file_bytes()
    t_frame = (time.perf_counter() - t0) / 10
    t0 = time.perf_counter()
    ref = _imageio.png_bytes(big)
    t_zlib = time.perf_counter() - t0
    print("3840x2160 result frame (2x of a synthetic 1080p frame), %.1f MB raw" % (big.size / 1e6))
    print("  GPU encoder, frame from host memory (H2D + kernel + framing): %6.1f ms, file %.2f MB" % (t_gpu * 1e3, len(png) / 1e6))
    print("  ... of which host framing (concatenate, Adler, CRC-32):        %6.1f ms" % (t_frame * 1e3))
    print("  zlib level 1 / Z_RLE / Sub on one core (cv2.imwrite defaults):  %6.1f ms, file %.2f MB" % (t_zlib * 1e3, len(ref) / 1e6))
    from PIL import Image
    import io
    assert (np.asarray(Image.open(io.BytesIO(png)))[:, :, ::-1] == big).all()


if __name__ == "__main__":
    n = int(sys.argv[1]) if len(sys.argv) > 1 else 240
    encoder_alone()
    sweep = os.path.join(ROOT, "tools", "png_route_sweep.py")
    for gpu_png in ("0", "1"):
        print("UVA_GPU_PNG=%s (%s)" % (gpu_png, "deflate on the GPU" if gpu_png == "1" else "zlib on the host"), flush=True)
        for g, et, dt in (("0", 8, 4), ("0", 8, 8), ("0,0", 4, 4), ("0,0", 8, 8)):
            env = dict(os.environ, UVA_GPU_PNG=gpu_png, UVA_ENCODE_THREADS=str(et), UVA_DECODE_THREADS=str(dt))
            subprocess.run([sys.executable, sweep, "--one", g, str(n)], env=env)
