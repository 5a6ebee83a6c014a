import os, sys
sys.path.insert(0, os.environ.get("GRAFT_REPO_ROOT", "/root/repo"))
import numpy as np
from upscale_video_amd import ncnn
from upscale_video_amd.synth import synthetic_frame
small = synthetic_frame(1080, 1920, seed=1)
big = np.repeat(np.repeat(small, 2, 0), 2, 1)
ws = ncnn.PngWorkspace(2160, 3840)
for _ in range(20): ncnn.png_encode_u8(big, workspace=ws)
