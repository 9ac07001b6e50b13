"""ctypes binding of include/ksa.h.  There is no fallback: if libksa.so cannot be loaded the
import raises, and every call that returns non-zero raises KsaError with ksa_last_error()."""
import ctypes as C
import os

HERE = os.path.dirname(os.path.abspath(__file__))
# The in-tree build, and nothing else: no environment variable selects another library or another kernel
# (A/B runs of variant builds swap the file: tools/with_lib.sh).
LIB_PATH = os.path.join(HERE, "libksa.so")

ABI_VERSION = 4
HM_ROWS = 128
CUMU = {"RAW": 0, "AVG": 1, "MAX": 2, "MIN": 3}
FMT_C64, FMT_U8 = 0, 1
OUT_LINEAR, OUT_DB, OUT_DB_CLIP = 0, 1, 2


class KsaError(RuntimeError):
    pass


class Config(C.Structure):
    _fields_ = [
        ("abi_version", C.c_int32), ("device", C.c_int32), ("fft_size", C.c_int32),
        ("full_size", C.c_int32), ("num_windows", C.c_int32),
        ("window_starts", C.POINTER(C.c_int32)), ("window", C.POINTER(C.c_float)),
        ("mag_scale", C.c_double), ("cumu_mode", C.c_int32), ("gain", C.c_float),
        ("min_amp", C.c_float), ("hm_width", C.c_int32), ("max_frames", C.c_int32),
        ("u8_offset", C.c_float), ("u8_scale", C.c_float),
        ("scan_total_entries", C.c_int32), ("scan_hop", C.c_int32), ("scan_hm_width", C.c_int32),
    ]


_P = C.c_void_p
_F = C.POINTER(C.c_float)
_I32, _I64 = C.c_int32, C.c_int64

# name -> (restype, argtypes); every symbol include/ksa.h declares
SIGNATURES = {
    "ksa_abi_version": (C.c_int, []),
    "ksa_last_error": (C.c_char_p, []),
    "ksa_create": (C.c_int, [C.POINTER(Config), C.POINTER(_P)]),
    "ksa_destroy": (None, [_P]),
    "ksa_set_stream": (C.c_int, [_P, _P]),
    "ksa_synchronize": (C.c_int, [_P]),
    "ksa_curscan_c64": (C.c_int, [_P, _P, _P]),
    "ksa_curscan_u8": (C.c_int, [_P, _P, _P]),
    "ksa_curscan_dev": (C.c_int, [_P, _P, _I32, _I64, _I32, _I32, _P]),
    "ksa_frames_dev": (C.c_int, [_P, _P, _I32, _I64, _I32, _I64, _I64, _P, _P, _I32]),
    "ksa_frame_c64": (C.c_int, [_P, _P]),
    "ksa_frame_u8": (C.c_int, [_P, _P]),
    "ksa_frame_spectrum": (C.c_int, [_P, _P]),
    "ksa_partial_dev": (C.c_int, [_P, C.POINTER(_P)]),
    "ksa_commit": (C.c_int, [_P, _I64]),
    "ksa_exchange_dev": (C.c_int, [_P, C.POINTER(_P), C.POINTER(_I64)]),
    "ksa_merge_gathered_dev": (C.c_int, [_P, _P, _I32, _I32, _I32]),
    "ksa_set_flags": (C.c_int, [_P, _I32, _I32, _I32]),
    "ksa_allreduce_state": (C.c_int, [C.POINTER(_P), _I32, _I32, _I32]),
    "ksa_set_adj": (C.c_int, [_P, _I32, _P, _I32]),
    "ksa_reset_state": (C.c_int, [_P]),
    "ksa_read_state": (C.c_int, [_P, _P, _P, _P, _P, _P, C.POINTER(_I32), C.POINTER(_I64)]),
    "ksa_state_dev": (C.c_int, [_P, C.POINTER(_P), C.POINTER(_P)]),
    "ksa_set_hm_index": (C.c_int, [_P, _I32]),
    "ksa_scan_pass_dev": (C.c_int, [_P, _P, _I32, _I64, _I32, _P]),
    "ksa_scan_pass_c64": (C.c_int, [_P, _P, _I32, _P]),
    "ksa_scan_pass_u8": (C.c_int, [_P, _P, _I32, _P]),
    "ksa_scan_stitch_dev": (C.c_int, [_P, _P, _I32]),
    "ksa_scan_passes_dev": (C.c_int, [_P, _P, _I32, _I64, _I32, _I32, _P]),
    "ksa_scan_stitch_passes_dev": (C.c_int, [_P, _P, _I32, _I32]),
    "ksa_scan_spectra_dev": (C.c_int, [_P, _P, _I32, _I64, _I32, _P, _P]),
    "ksa_scan_stitch_range_dev": (C.c_int, [_P, _P, _I32, _P, _I32, _I32, _I32, _I32, _I32, _I32, _I32]),
    "ksa_scan_rows_dev": (C.c_int, [_P, C.POINTER(_P), C.POINTER(_I32)]),
    "ksa_scan_merge_rows_dev": (C.c_int, [_P, _P, _I32, _I32, _I32]),
    "ksa_scan_allstitch": (C.c_int, [C.POINTER(_P), _I32, C.POINTER(_P), _I32, _I32]),
    "ksa_scan_gather_state": (C.c_int, [C.POINTER(_P), _I32, _I32, _P, _P, _P, _P]),
    "ksa_scan_read_state": (C.c_int, [_P, _P, _P, _P, _P, _P, C.POINTER(_I32), C.POINTER(_I64)]),
    "ksa_scan_state_dev": (C.c_int, [_P, C.POINTER(_P), C.POINTER(_P)]),
    "ksa_scan_reset": (C.c_int, [_P]),
    "ksa_scan_set_base_is_raw": (C.c_int, [_P, _I32]),
    "ksa_read_levels": (C.c_int, [_P, _I32, _I32, _I32, _P]),
    "ksa_read_highs": (C.c_int, [_P, _I32, _I32, _I32, _I32, C.c_double, _I32, _P, _P, C.POINTER(_I32)]),
    "ksa_read_hm_rows": (C.c_int, [_P, _I32, _I32, _I32, _P]),
    "ksa_read_view": (C.c_int, [_P, _I32, _I32, _I32, _P, _I32, C.c_double, _I32, _P, _P, C.POINTER(_I32), _I32, _P, C.POINTER(_I32)]),
    "ksa_host_alloc": (C.c_int, [C.POINTER(_P), _I64]),
    "ksa_host_free": (C.c_int, [_P]),
    "ksa_prof_enable": (C.c_int, [_P, _I32]),
    "ksa_prof_read": (C.c_int, [_P, C.POINTER(C.c_double), C.POINTER(_I64)]),
    "ksa_prof_clock": (C.c_int, [_P, C.POINTER(C.c_double), C.POINTER(C.c_double), C.POINTER(C.c_double), C.POINTER(_I64)]),
    "ksa_kernel_info": (C.c_int, [_P] + [C.POINTER(_I32)] * 5),
}


def _preload_torch_hip_runtime():
    """PyTorch-ROCm wheels bundle their own libamdhip64.so (same SONAME as /opt/rocm's).  Two HIP
    runtimes in one process do not both see the GPU, so when torch is installed its runtime is mapped
    first; libksa's DT_NEEDED libamdhip64.so.7 then binds to that one copy (streams, device memory and
    RCCL are shared with torch).  Without torch the system runtime under /opt/rocm is used."""
    import importlib.util
    try:
        spec = importlib.util.find_spec("torch")
    except (ImportError, ValueError):
        spec = None
    if spec is None or not spec.submodule_search_locations:
        return None
    cand = os.path.join(list(spec.submodule_search_locations)[0], "lib", "libamdhip64.so")
    if os.path.exists(cand):
        C.CDLL(cand, mode=C.RTLD_GLOBAL)
        return cand
    return None


def load(path=LIB_PATH):
    _preload_torch_hip_runtime()
    if not os.path.exists(path):
        raise KsaError("libksa.so is missing at %s -- build it with `python __graft_entry__.py` "
                       "(hipcc --offload-arch=gfx950); there is no CPU fallback" % path)
    lib = C.CDLL(path)
    lib.ksa_abi_version.restype = C.c_int
    if lib.ksa_abi_version() != ABI_VERSION:      # checked first: a stale build is named as such, not as a missing symbol
        raise KsaError("%s has ABI %d, this binding expects %d -- rebuild it (python __graft_entry__.py)"
                       % (path, lib.ksa_abi_version(), ABI_VERSION))
    for name, (res, args) in SIGNATURES.items():
        fn = getattr(lib, name)   # AttributeError if the library does not export it
        fn.restype = res
        fn.argtypes = args
    return lib


lib = load()


def check(rc):
    if rc != 0:
        raise KsaError(lib.ksa_last_error().decode("utf-8", "replace"))
