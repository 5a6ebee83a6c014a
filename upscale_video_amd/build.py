"""Builds libuva.so (hand-written HIP kernels + C ABI) in-tree for gfx950 with hipcc."""
import os
import shutil
import subprocess

_HERE = os.path.dirname(os.path.abspath(__file__))
CSRC = os.path.join(_HERE, "csrc")
LIB = os.path.join(_HERE, "libuva.so")
SOURCES = ["uva_api.hip", "uva_model.cpp", "uva_generic.cpp", "uva_pngread.cpp"]
DEPS = SOURCES + ["uva_kernels.hip.h", "uva_rdb.hip.h", "uva_generic.hip.h", "uva_generic.h", "uva_model.h", "uva_png.hip.h", "uva_denoise.hip.h", os.path.join("..", "..", "include", "uva.h")]


def hipcc():
    exe = shutil.which("hipcc") or "/opt/rocm/bin/hipcc"
    if not os.path.exists(exe):
        raise RuntimeError("hipcc not found: libuva.so cannot be built")
    return exe


def needs_build():
    if not os.path.exists(LIB):
        return True
    t = os.path.getmtime(LIB)
    return any(os.path.getmtime(os.path.join(CSRC, d)) > t for d in DEPS)


def build_instrumented(verbose=False):
    """libuva_instr.so: the same sources with -DUVA_INSTRUMENT (in-kernel cycle stamps, the trunk kernel's
    ablation variants, uva_net_debug_trunk_stamps).  Select it with UVA_LIB_PATH; tools/ only."""
    out = os.path.join(_HERE, "libuva_instr.so")
    cmd = [hipcc(), "--offload-arch=gfx950", "-O3", "-std=c++17", "-fPIC", "-shared", "-DUVA_INSTRUMENT",
           "-Wall", "-Wno-unused-function"] + [os.path.join(CSRC, s) for s in SOURCES] + ["-o", out]
    if verbose:
        print(" ".join(cmd))
    subprocess.check_call(cmd)
    return out


def build_lib(force=False, verbose=False):
    """hipcc --offload-arch=gfx950 (cross-compiles without a GPU).  Returns the .so path."""
    if os.environ.get("UVA_LIB_PATH"):      # an A/B build selected by the caller: leave it alone
        return os.environ["UVA_LIB_PATH"]
    if not force and not needs_build():
        return LIB
    # one builder at a time (bench.py runs one process per GPU): the others wait, then find it fresh
    import fcntl
    with open(LIB + ".lock", "w") as lock:
        fcntl.flock(lock, fcntl.LOCK_EX)
        if not force and not needs_build():
            return LIB
        tmp = "%s.tmp.%d" % (LIB, os.getpid())
        cmd = [hipcc(), "--offload-arch=gfx950", "-O3", "-std=c++17", "-fPIC", "-shared",
               "-Wall", "-Wno-unused-function"]
        cmd += [os.path.join(CSRC, s) for s in SOURCES]
        cmd += ["-o", tmp]
        if verbose:
            print(" ".join(cmd))
        try:
            subprocess.check_call(cmd)
            os.replace(tmp, LIB)
        finally:
            if os.path.exists(tmp):
                os.remove(tmp)
    return LIB


if __name__ == "__main__":
    import sys
    if "--instrument" in sys.argv:
        print(build_instrumented(verbose=True))
    else:
        print(build_lib(force=True, verbose=True))
