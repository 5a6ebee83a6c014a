"""Builds libuva.so (hand-written HIP kernels + C ABI) in-tree for gfx950 with hipcc.

Every source file is compiled to an object of its own (in parallel; objects are kept under csrc/_obj/ and only
rebuilt when the source or one of the headers it includes changed) and the objects are linked into the library:
the kernel files take minutes, a change to one of them should not cost all of them."""
import os
import shutil
import subprocess
from concurrent.futures import ThreadPoolExecutor

_HERE = os.path.dirname(os.path.abspath(__file__))
CSRC = os.path.join(_HERE, "csrc")
LIB = os.path.join(_HERE, "libuva.so")
_UVA_H = os.path.join("..", "..", "include", "uva.h")
# source -> the headers it includes (directly or not)
SOURCES = {
    "uva_api.hip": ["uva_kernels.hip.h", "uva_devutil.hip.h", "uva_wino.h", "uva_sub5.h", "uva_sub10.h", "uva_sw.h", "uva_rdb.hip.h", "uva_generic.hip.h", "uva_generic.h",
                    "uva_model.h", "uva_png.hip.h", "uva_denoise.hip.h", _UVA_H],
    "uva_wino.hip": ["uva_wino.hip.h", "uva_wino.h", "uva_devutil.hip.h"],
    "uva_sub5.hip": ["uva_sub5.hip.h", "uva_sub5.h", "uva_devutil.hip.h", "uva_model.h"],
    "uva_sub10.hip": ["uva_sub10.hip.h", "uva_sub10.h", "uva_devutil.hip.h", "uva_model.h"],
    "uva_sww.hip": ["uva_sww.hip.h", "uva_sw.h", "uva_devutil.hip.h"],
    "uva_model.cpp": ["uva_model.h"],
    "uva_generic.cpp": ["uva_generic.h", "uva_model.h"],
    "uva_pngread.cpp": [_UVA_H],
}
FLAGS = ["--offload-arch=gfx950", "-O3", "-std=c++17", "-fPIC", "-Wall", "-Wno-unused-function"]


def hipcc():
    exe = shutil.which("hipcc") or "/opt/rocm/bin/hipcc"
    if not os.path.exists(exe):
        raise RuntimeError("hipcc not found: libuva.so cannot be built")
    return exe


def _mtime(name):
    return os.path.getmtime(os.path.join(CSRC, name))


def _stale(obj, src):
    if not os.path.exists(obj):
        return True
    t = os.path.getmtime(obj)
    return any(_mtime(d) > t for d in [src] + SOURCES[src])


def _obj(objdir, src):
    return os.path.join(objdir, src.rsplit(".", 1)[0] + ".o")


def needs_build():
    if not os.path.exists(LIB):
        return True
    t = os.path.getmtime(LIB)
    if any(_mtime(d) > t for src, deps in SOURCES.items() for d in [src] + deps):
        return True
    # a header edited while a build was running: the library is newer than the header, the object that includes it is not
    objdir = os.path.join(CSRC, "_obj")
    if not os.path.isdir(objdir):
        return False                    # a library shipped without its objects (the GPU box): nothing to compare
    return any(os.path.exists(_obj(objdir, src)) and (_stale(_obj(objdir, src), src) or os.path.getmtime(_obj(objdir, src)) > t)
               for src in SOURCES)


def _build(out, objdir, defines, force, verbose):
    os.makedirs(objdir, exist_ok=True)
    todo = []
    objs = []
    for src in SOURCES:
        obj = _obj(objdir, src)
        objs.append(obj)
        if force or _stale(obj, src):
            todo.append((src, obj))

    def compile_one(job):
        src, obj = job
        tmp = "%s.tmp.%d" % (obj, os.getpid())
        cmd = [hipcc()] + FLAGS + defines + ["-c", os.path.join(CSRC, src), "-o", tmp]
        if verbose:
            print(" ".join(cmd))
        try:
            subprocess.check_call(cmd)
            os.replace(tmp, obj)
        finally:
            if os.path.exists(tmp):
                os.remove(tmp)
    with ThreadPoolExecutor(max_workers=max(1, min(len(todo), os.cpu_count() or 1))) as ex:
        list(ex.map(compile_one, todo))
    tmp = "%s.tmp.%d" % (out, os.getpid())
    cmd = [hipcc(), "--offload-arch=gfx950", "-shared", "-fPIC"] + objs + ["-o", tmp]
    if verbose:
        print(" ".join(cmd))
    try:
        subprocess.check_call(cmd)
        os.replace(tmp, out)
    finally:
        if os.path.exists(tmp):
            os.remove(tmp)
    return out


def build_instrumented(verbose=False):
    """libuva_instr.so: the same sources with -DUVA_INSTRUMENT (in-kernel cycle stamps, the trunk kernel's
    ablation variants, uva_net_debug_trunk_stamps).  Select it with UVA_LIB_PATH; tools/ only."""
    return _build(os.path.join(_HERE, "libuva_instr.so"), os.path.join(CSRC, "_obj_instr"), ["-DUVA_INSTRUMENT"], False, verbose)


def build_variant(name, defines, verbose=False):
    """libuva_<name>.so with extra -D defines (A/B builds; select with UVA_LIB_PATH)."""
    return _build(os.path.join(_HERE, "libuva_%s.so" % name), os.path.join(CSRC, "_obj_" + name), list(defines), False, verbose)


def rebuild_trunkw(out, defines=(), verbose=False):
    """Compile csrc/uva_wino.hip (trunkw_kernel: the headline kernel, its own translation unit, seconds) HERE and link it with
    the other objects of libuva.so as they lie under csrc/_obj/ -> `out`.  What `pytest -m gpu` uses to show that the kernel the
    box measures also COMPILES on the box (tests/test_gpu_workers.py); objects that are missing are compiled too (minutes)."""
    import tempfile
    objdir = os.path.join(CSRC, "_obj")
    os.makedirs(objdir, exist_ok=True)
    with tempfile.TemporaryDirectory(prefix="uva_rebuild_") as tmp:
        objs = []
        for src in SOURCES:
            obj = _obj(objdir, src)
            if src == "uva_wino.hip" or _stale(obj, src):
                # (the kernel's object stays in the scratch directory; a stale object of another file is rebuilt for good, through
                # a temporary name like _build's, so that a concurrent build never links half a file)
                dst = os.path.join(tmp, os.path.basename(obj))
                cmd = [hipcc()] + FLAGS + list(defines if src == "uva_wino.hip" else []) + ["-c", os.path.join(CSRC, src), "-o", dst]
                if verbose:
                    print(" ".join(cmd))
                subprocess.check_call(cmd)
                if src != "uva_wino.hip":
                    keep = "%s.tmp.%d" % (obj, os.getpid())
                    shutil.copyfile(dst, keep)
                    os.replace(keep, obj)
                obj = dst
            objs.append(obj)
        subprocess.check_call([hipcc(), "--offload-arch=gfx950", "-shared", "-fPIC"] + objs + ["-o", out])
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
        _build(LIB, os.path.join(CSRC, "_obj"), [], force, verbose)
    return LIB


if __name__ == "__main__":
    import sys
    if "--instrument" in sys.argv:
        print(build_instrumented(verbose=True))
    elif "--variant" in sys.argv:
        i = sys.argv.index("--variant")
        print(build_variant(sys.argv[i + 1], sys.argv[i + 2:], verbose=True))
    else:
        print(build_lib(force="--force" in sys.argv, verbose=True))
