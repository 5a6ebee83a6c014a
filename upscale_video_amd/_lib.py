"""ctypes binding of libuva.so (C ABI: include/uva.h).  Fails loudly: there is no fallback path."""
import ctypes
import os

_HERE = os.path.dirname(os.path.abspath(__file__))
LIB_PATH = os.environ.get("UVA_LIB_PATH") or os.path.join(_HERE, "libuva.so")   # override: A/B builds

# every symbol include/uva.h declares
ABI_VERSION = 8   # include/uva.h UVA_ABI_VERSION

SYMBOLS = [
    "uva_get_gpu_count", "uva_get_default_gpu_index", "uva_get_gpu_info", "uva_get_gpu_pci_bus_id",
    "uva_debug_trunk2_schedule", "uva_debug_sub10_rows", "uva_net_submit_u8_png", "uva_png_workspace_bytes",
    "uva_png_assemble", "uva_png_deflate_u8", "uva_debug_png_deflate_host", "uva_png_decode_bgr", "uva_debug_zlib_decompress", "uva_net_debug_generic_plan", "uva_denoise_u8", "uva_debug_denoise_stage", "uva_destroy_gpu_instance",
    "uva_net_create", "uva_net_set_device", "uva_net_load_param", "uva_net_load_model",
    "uva_net_destroy", "uva_net_scale", "uva_net_num_features", "uva_net_num_convs",
    "uva_net_extract_f32", "uva_net_process_u8", "uva_net_process_u8_device", "uva_net_synchronize",
    "uva_net_wait_for", "uva_net_submit_u8", "uva_net_collect_u8", "uva_host_alloc", "uva_host_free",
    "uva_net_debug_read_activation", "uva_net_set_profiling", "uva_net_kernel_stats",
    "uva_net_debug_packed_weights", "uva_last_error", "uva_abi_version",
]
INSTRUMENT_SYMBOLS = ["uva_net_debug_trunk_stamps"]     # only in a -DUVA_INSTRUMENT build (build.py --instrument)

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
    L.uva_get_gpu_count.restype = c_i
    L.uva_get_default_gpu_index.restype = c_i
    L.uva_get_gpu_info.restype = c_i
    L.uva_get_gpu_info.argtypes = [c_i, ctypes.POINTER(c_i), ctypes.c_char_p, c_sz]
    if not hasattr(L, "uva_get_gpu_pci_bus_id"):     # an older A/B build (UVA_LIB_PATH)
        _lib = L
        return L
    L.uva_get_gpu_pci_bus_id.argtypes = [c_i, ctypes.c_char_p, c_sz]
    if hasattr(L, "uva_png_assemble"):
        L.uva_net_submit_u8_png.restype = ctypes.c_longlong
        L.uva_net_submit_u8_png.argtypes = [c_p, c_p, c_i, c_i, c_sz, c_p, c_sz, c_i, c_i]
        L.uva_png_workspace_bytes.restype = c_sz
        L.uva_png_workspace_bytes.argtypes = [c_i, c_i]
        L.uva_png_assemble.argtypes = [c_p, c_i, c_i, c_p, c_sz, ctypes.POINTER(c_sz)]
        L.uva_png_deflate_u8.argtypes = [c_i, c_p, c_i, c_i, c_sz, c_p, c_sz]
        L.uva_debug_png_deflate_host.argtypes = [c_p, c_i, c_i, c_sz, c_p, c_sz]
    if hasattr(L, "uva_net_debug_generic_plan"):
        L.uva_net_debug_generic_plan.argtypes = [c_p, ctypes.POINTER(c_i)]
    if hasattr(L, "uva_png_decode_bgr"):
        L.uva_png_decode_bgr.argtypes = [c_p, c_sz, c_p, c_sz, ctypes.POINTER(c_i), ctypes.POINTER(c_i)]
        L.uva_debug_zlib_decompress.argtypes = [c_p, c_sz, c_p, c_sz]
    if hasattr(L, "uva_debug_sub10_rows"):
        L.uva_debug_sub10_rows.argtypes = [c_i, c_i, c_i, c_p, c_sz, ctypes.POINTER(c_sz), c_p, ctypes.POINTER(c_i)]
    L.uva_debug_trunk2_schedule.argtypes = [c_i, c_i, c_i, c_i, c_i, c_p, c_sz, ctypes.POINTER(c_sz), c_p,
                                            ctypes.POINTER(c_i), c_p, c_i, ctypes.POINTER(c_i),
                                            ctypes.POINTER(ctypes.c_longlong)]
    L.uva_denoise_u8.argtypes = [c_i, c_p, c_i, c_i, c_sz, c_p, c_sz, ctypes.c_float, ctypes.c_float]
    L.uva_debug_denoise_stage.argtypes = [c_i, c_i, c_p, c_i, c_i, ctypes.c_float, c_p]
    L.uva_destroy_gpu_instance.restype = None
    L.uva_net_create.restype = c_p
    L.uva_net_destroy.argtypes = [c_p]
    L.uva_net_destroy.restype = None
    L.uva_net_set_device.argtypes = [c_p, c_i]
    L.uva_net_load_param.argtypes = [c_p, ctypes.c_char_p]
    L.uva_net_load_model.argtypes = [c_p, ctypes.c_char_p]
    for n in ("uva_net_scale", "uva_net_num_features", "uva_net_num_convs", "uva_net_synchronize"):
        getattr(L, n).argtypes = [c_p]
    L.uva_net_wait_for.argtypes = [c_p, c_p]
    L.uva_net_extract_f32.argtypes = [c_p, c_p, c_i, c_i, c_p]
    L.uva_net_process_u8.argtypes = [c_p, c_p, c_i, c_i, c_sz, c_p, c_sz, c_i, c_i]
    L.uva_net_submit_u8.argtypes = [c_p, c_p, c_i, c_i, c_sz, c_p, c_sz, c_i, c_i]
    L.uva_net_submit_u8.restype = ctypes.c_longlong
    L.uva_net_collect_u8.argtypes = [c_p, ctypes.c_longlong]
    L.uva_host_alloc.argtypes = [c_sz]
    L.uva_host_alloc.restype = c_p
    L.uva_host_free.argtypes = [c_p]
    L.uva_host_free.restype = None
    L.uva_net_process_u8_device.argtypes = [c_p, c_p, c_i, c_i, c_sz, c_p, c_sz, c_i, c_i]
    L.uva_net_debug_read_activation.argtypes = [c_p, c_i, c_p, c_i, c_i]
    L.uva_net_set_profiling.argtypes = [c_p, c_i]
    L.uva_net_kernel_stats.argtypes = [c_p, c_i, ctypes.POINTER(ctypes.c_longlong),
                                       ctypes.POINTER(ctypes.c_double)]
    L.uva_net_debug_packed_weights.argtypes = [c_p, c_i, c_p, c_sz, ctypes.POINTER(c_sz)]
    if hasattr(L, "uva_net_debug_trunk_stamps"):
        L.uva_net_debug_trunk_stamps.argtypes = [c_p, c_p, c_i, ctypes.POINTER(c_i), c_i, ctypes.POINTER(ctypes.c_float)]
    L.uva_last_error.restype = ctypes.c_char_p
    L.uva_abi_version.restype = c_i
    ab_build = bool(os.environ.get("UVA_LIB_PATH"))   # an older build under comparison may lack newer entry points
    for n in SYMBOLS:   # AttributeError here means the .so is stale: rebuild it
        if not ab_build:
            getattr(L, n)
    if L.uva_abi_version() != ABI_VERSION and not ab_build:
        raise UvaError("libuva.so has ABI version %d, this package needs %d: rebuild it" % (L.uva_abi_version(), ABI_VERSION))
    _lib = L
    return L


def check(rc):
    if rc != 0:
        raise UvaError(load().uva_last_error().decode(errors="replace"))
