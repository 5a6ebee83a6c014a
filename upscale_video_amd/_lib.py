"""ctypes binding of libuva.so (C ABI: include/uva.h).  Fails loudly: there is no fallback path."""
import ctypes
import os

_HERE = os.path.dirname(os.path.abspath(__file__))
LIB_PATH = os.environ.get("UVA_LIB_PATH") or os.path.join(_HERE, "libuva.so")   # override: A/B builds

# every symbol include/uva.h declares
ABI_VERSION = 15  # include/uva.h UVA_ABI_VERSION

SYMBOLS = [
    "uva_get_gpu_count", "uva_get_default_gpu_index", "uva_get_gpu_info", "uva_get_gpu_pci_bus_id",
    "uva_debug_trunk2_schedule", "uva_debug_trunkw_schedule", "uva_debug_sub10_rows", "uva_debug_sub10_rows_batch", "uva_debug_sub5_rows", "uva_net_submit_u8_png", "uva_png_workspace_bytes",
    "uva_png_assemble", "uva_png_deflate_u8", "uva_debug_png_deflate_host", "uva_png_decode_bgr", "uva_debug_zlib_decompress", "uva_net_debug_generic_plan", "uva_debug_generic_segments", "uva_debug_generic_segments_planes", "uva_debug_generic_batches", "uva_denoise_u8", "uva_denoise_u8_device", "uva_denoise_synchronize", "uva_debug_denoise_stage", "uva_destroy_gpu_instance",
    "uva_net_create", "uva_net_set_device", "uva_net_device", "uva_net_load_param", "uva_net_load_model",
    "uva_net_destroy", "uva_net_scale", "uva_net_num_features", "uva_net_num_convs",
    "uva_net_extract_f32", "uva_net_process_u8", "uva_net_process_u8_device", "uva_net_process_u8_device_batch", "uva_net_synchronize",
    "uva_net_wait_for", "uva_net_submit_u8", "uva_net_collect_u8", "uva_host_alloc", "uva_host_free",
    "uva_net_debug_read_activation", "uva_net_set_profiling", "uva_net_kernel_stats",
    "uva_net_debug_packed_weights", "uva_last_error", "uva_abi_version",
]
INSTRUMENT_SYMBOLS = ["uva_net_debug_trunk_stamps", "uva_net_debug_rdb_stamps"]     # only in a -DUVA_INSTRUMENT build (build.py --instrument)

_lib = None


class UvaError(RuntimeError):
    pass


def load():
    """Loads libuva.so; raises if it has not been built (python -m upscale_video_amd.build)."""
    global _lib
    if _lib is not None:
        return _lib
    if not os.path.exists(LIB_PATH):
        raise UvaError(
            "libuva.so is missing: build the HIP extension first "
            "(python -c 'import __graft_entry__ as g; g.build()' or python upscale_video_amd/build.py). "
            "There is no CPU fallback.")
    L = ctypes.CDLL(LIB_PATH)
    c_p, c_i, c_sz = ctypes.c_void_p, ctypes.c_int, ctypes.c_size_t
    # The ABI check comes first and applies to UVA_LIB_PATH builds too (that variable is the documented way to select a
    # library: a stale one would be called with today's argument lists).  An older build under A/B comparison is let
    # through only by the explicit UVA_ALLOW_OLD_ABI=1; entry points it lacks are then simply not declared.
    if not hasattr(L, "uva_abi_version"):
        raise UvaError("%s is not a libuva build (no uva_abi_version)" % LIB_PATH)
    L.uva_abi_version.restype = c_i
    old_ok = os.environ.get("UVA_ALLOW_OLD_ABI") == "1"
    if L.uva_abi_version() != ABI_VERSION and not old_ok:
        raise UvaError("libuva.so has ABI version %d, this package needs %d: rebuild it (UVA_ALLOW_OLD_ABI=1 lets an older "
                       "A/B build through)" % (L.uva_abi_version(), ABI_VERSION))
    if not old_ok:
        for n in SYMBOLS:   # AttributeError here means the .so is stale: rebuild it
            getattr(L, n)

    def decl(name, argtypes=None, restype=None, keep_restype=False):
        if not hasattr(L, name):
            return
        f = getattr(L, name)
        if argtypes is not None:
            f.argtypes = argtypes
        if not keep_restype:
            f.restype = c_i if restype is None else (None if restype == "void" else restype)

    pi, psz, pll = ctypes.POINTER(c_i), ctypes.POINTER(c_sz), ctypes.POINTER(ctypes.c_longlong)
    # pointer- and wide-valued results first: the default int restype would truncate them
    decl("uva_net_create", [], c_p)
    decl("uva_host_alloc", [c_sz], c_p)
    decl("uva_last_error", [], ctypes.c_char_p)
    decl("uva_net_submit_u8", [c_p, c_p, c_i, c_i, c_sz, c_p, c_sz, c_i, c_i], ctypes.c_longlong)
    decl("uva_net_submit_u8_png", [c_p, c_p, c_i, c_i, c_sz, c_p, c_sz, c_i, c_i], ctypes.c_longlong)
    decl("uva_png_workspace_bytes", [c_i, c_i], c_sz)
    decl("uva_get_gpu_count", [])
    decl("uva_get_default_gpu_index", [])
    decl("uva_get_gpu_info", [c_i, pi, ctypes.c_char_p, c_sz])
    decl("uva_get_gpu_pci_bus_id", [c_i, ctypes.c_char_p, c_sz])
    decl("uva_png_assemble", [c_p, c_i, c_i, c_p, c_sz, psz])
    decl("uva_png_deflate_u8", [c_i, c_p, c_i, c_i, c_sz, c_p, c_sz])
    decl("uva_debug_png_deflate_host", [c_p, c_i, c_i, c_sz, c_p, c_sz])
    decl("uva_net_debug_generic_plan", [c_p, pi])
    decl("uva_debug_generic_segments", [c_i, c_i, c_i, c_i, c_p, c_sz, psz, c_p])
    decl("uva_debug_generic_segments_planes", [c_i, c_p, c_i, c_i, c_p, c_sz, psz, c_p])
    decl("uva_debug_generic_batches", [c_i, c_i, c_i, c_i, ctypes.c_longlong, c_p, c_sz, psz])
    decl("uva_png_decode_bgr", [c_p, c_sz, c_p, c_sz, pi, pi])
    decl("uva_debug_zlib_decompress", [c_p, c_sz, c_p, c_sz])
    decl("uva_debug_sub10_rows", [c_i, c_i, c_i, c_p, c_sz, psz, c_p, pi])
    decl("uva_debug_sub10_rows_batch", [c_i, c_i, c_i, c_i, c_p, c_sz, psz, c_p, pi])
    decl("uva_debug_sub5_rows", [c_i, c_i, c_i, c_p, c_sz, psz, c_p, pi])
    decl("uva_debug_trunk2_schedule", [c_i, c_i, c_i, c_i, c_i, c_p, c_sz, psz, c_p, pi, c_p, c_i, pi, pll])
    decl("uva_debug_trunkw_schedule", [c_i, c_i, c_i, c_i, c_i, c_p, c_sz, psz, c_p, pi, c_p, c_i, pi, pll])
    decl("uva_denoise_u8", [c_i, c_p, c_i, c_i, c_sz, c_p, c_sz, ctypes.c_float, ctypes.c_float])
    decl("uva_denoise_u8_device", [c_i, c_p, c_i, c_i, c_sz, c_p, c_sz, ctypes.c_float, ctypes.c_float, c_p, c_p])
    decl("uva_denoise_synchronize", [c_i])
    decl("uva_debug_denoise_stage", [c_i, c_i, c_p, c_i, c_i, ctypes.c_float, c_p])
    decl("uva_destroy_gpu_instance", [], "void")
    decl("uva_net_destroy", [c_p], "void")
    decl("uva_net_set_device", [c_p, c_i])
    decl("uva_net_load_param", [c_p, ctypes.c_char_p])
    decl("uva_net_load_model", [c_p, ctypes.c_char_p])
    for n in ("uva_net_scale", "uva_net_num_features", "uva_net_num_convs", "uva_net_synchronize", "uva_net_device"):
        decl(n, [c_p])
    decl("uva_net_wait_for", [c_p, c_p])
    decl("uva_net_extract_f32", [c_p, c_p, c_i, c_i, c_p])
    decl("uva_net_process_u8", [c_p, c_p, c_i, c_i, c_sz, c_p, c_sz, c_i, c_i])
    decl("uva_net_collect_u8", [c_p, ctypes.c_longlong])
    decl("uva_host_free", [c_p], "void")
    decl("uva_net_process_u8_device", [c_p, c_p, c_i, c_i, c_sz, c_p, c_sz, c_i, c_i])
    decl("uva_net_process_u8_device_batch", [c_p, c_p, c_p, c_i, c_i, c_i, c_sz, c_sz, c_i, c_i])
    decl("uva_net_debug_read_activation", [c_p, c_i, c_p, c_i, c_i])
    decl("uva_net_set_profiling", [c_p, c_i])
    decl("uva_net_kernel_stats", [c_p, c_i, pll, ctypes.POINTER(ctypes.c_double)])
    decl("uva_net_debug_packed_weights", [c_p, c_i, c_p, c_sz, psz])
    decl("uva_net_debug_trunk_stamps", [c_p, c_p, c_i, pi, c_i, ctypes.POINTER(ctypes.c_float)])
    decl("uva_net_debug_rdb_stamps", [c_p, c_p, c_i])
    _lib = L
    return L


def check(rc):
    if rc != 0:
        raise UvaError(load().uva_last_error().decode(errors="replace"))
