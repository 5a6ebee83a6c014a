import os
import sys

import numpy as np
import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)

# libuva.so honours its A/B and debug switches (UVA_TRUNK_WINO, UVA_TW_FOLD, ...) only under this opt-in; the tests use them
# (csrc/uva_devutil.hip.h debug_env; tests/test_host.py::test_debug_switches_need_the_opt_in checks the gate itself)
os.environ["UVA_DEBUG_SWITCHES"] = "1"


def pytest_configure(config):
    config.addinivalue_line("markers", "gpu: needs a real MI355X (run with -m gpu on the GPU box)")


def pytest_sessionfinish(session, exitstatus):
    """the measured distances of every oracle comparison of this session -> gpurun_out/parity_report.json"""
    import parity_report
    parity_report.write()


def pytest_terminal_summary(terminalreporter, exitstatus, config):
    """... and one line each in the log, whatever the verbosity (the driver runs `-q`)"""
    import parity_report
    lines = list(parity_report.summary_lines())
    if lines:
        terminalreporter.write_sep("-", "measured parity (tests/parity_report.py; %d comparisons)" % len(lines))
        for ln in lines:
            terminalreporter.write_line(ln)


@pytest.fixture(scope="session", autouse=True)
def _torch_cuda_first(request):
    """Tests that hand torch CUDA tensors to the library need torch's own HIP runtime initialised BEFORE libuva.so brings up
    the one it links against: the other way round torch reports "No HIP GPUs are available" (seen with
    `pytest tests/test_gpu_workers.py tests/test_rawvideo.py -m gpu`; the whole suite happened to run in a good order)."""
    if any(item.get_closest_marker("gpu") for item in request.session.items):
        try:
            import torch
            if torch.cuda.is_available():
                torch.cuda.init()
        except Exception:
            pass
    yield


@pytest.fixture(scope="session")
def oracle():
    """The CPU oracle (test infrastructure; oracle/oracle.c built with gcc)."""
    from oracle import uvoracle
    uvoracle.build()
    return uvoracle


@pytest.fixture(scope="session")
def oracle_models(oracle):
    return {k: oracle.load_model(k) for k in ("2x", "4x", "1x")}


@pytest.fixture(scope="session")
def uva():
    """The product library; built on demand (hipcc cross-compiles without a GPU)."""
    from upscale_video_amd import build
    build.build_lib()
    from upscale_video_amd import ncnn
    return ncnn


def model_paths(key):
    from oracle import uvoracle
    base = os.path.join(ROOT, "models", uvoracle.MODEL_FILES[key])
    return base + ".param", base + ".bin"


def load_net(ncnn, key, device=0):
    net = ncnn.Net()
    net.opt.use_vulkan_compute = True
    net.set_vulkan_device(device)
    p, b = model_paths(key)
    assert net.load_param(p) == 0, getattr(net, "last_error", "")
    assert net.load_model(b) == 0, getattr(net, "last_error", "")
    return net


def psnr_u8(a, b):
    d = a.astype(np.float64) - b.astype(np.float64)
    mse = float((d * d).mean())
    return 99.0 if mse == 0 else 10 * np.log10(255.0 ** 2 / mse)
