"""ctypes binding of libcerberus_hip.so (the C ABI in include/cerberus_hip.h).

There is deliberately NO fallback: if the shared library has not been built
(``python -m cerberusnet_amd.build`` / ``__graft_entry__.build()``) every op
raises ``RuntimeError`` -- a GPU box must never silently run something else.
"""
import ctypes
import os
import threading

_HERE = os.path.dirname(os.path.abspath(__file__))
# CERBERUS_HIP_LIB: another build of the same library (e.g. a -DCERB_EXPERIMENTS / -DCERB_STAMP test build)
LIB_PATH = os.environ.get("CERBERUS_HIP_LIB") or os.path.join(_HERE, "lib", "libcerberus_hip.so")
ABI_VERSION = 7

_lock = threading.Lock()
_lib = None

_I = ctypes.c_int
_P = ctypes.c_void_p
_I64 = ctypes.c_int64

# name -> (restype, argtypes); must list every symbol include/cerberus_hip.h declares
PROTOTYPES = {
    "cerberus_abi_version": (_I, []),
    "cerberus_error_string": (ctypes.c_char_p, [_I]),
    "cerberus_correlation_out_shape": (_I, [_I] * 7 + [ctypes.POINTER(_I)] * 3),
    "cerberus_correlation_forward": (_I, [_P, _P, _P] + [_I] * 11 + [_P]),
    "cerberus_correlation_forward_ex": (_I, [_P, _P, _P] + [_I] * 9 +
                                        [ctypes.c_float, _I64, _I, _P]),
    "cerberus_correlation_backward": (_I, [_P, _P, _P, _P, _P] + [_I] * 11 + [_P]),
    "cerberus_correlation_backward_ex_workspace_bytes": (_I64, [_I] * 9),
    "cerberus_correlation_backward_ex": (_I, [_P, _P, _P, _I64, _P, _I64, ctypes.c_float, _P, _I64, _P, _P] + [_I] * 10 + [_P]),
    "cerberus_flow_warp_forward": (_I, [_P, _P, _P] + [_I] * 7 + [_P]),
    "cerberus_flow_warp_context_bytes": (_I64, [_I, _I, _I]),
    "cerberus_flow_warp_forward_ctx": (_I, [_P, _P, _P, _P, _I64] + [_I] * 8 + [_P]),
    "cerberus_warp_correlation_workspace_bytes": (_I64, [_I, _I, _I, _I]),
    "cerberus_warp_correlation_forward": (_I, [_P, _P, _P, _P, _P, _I64] + [_I] * 5 + [ctypes.c_float, _I64, _I, _I, _P]),
    "cerberus_flow_warp_backward_workspace_bytes": (_I64, [_I, _I, _I, _I]),
    "cerberus_flow_warp_backward": (_I, [_P, _P, _P, _P, _P, _P, _I64, _P, _I64] + [_I] * 8 + [_P]),
    "cerberus_flow_upsample_forward": (_I, [_P, _P, _I64, _I, _I, _I, _I, _P]),
    "cerberus_flow_upsample_backward": (_I, [_P, _P, _I64, _I, _I, _I, _I, _P]),
    "cerberus_area_resize": (_I, [_P, _P, _I64, _I, _I, _I, _I, _I, _P]),
    "cerberus_area_pyramid": (_I, [_P, ctypes.POINTER(_P), ctypes.POINTER(_I), ctypes.POINTER(_I), _I, _I64, _I, _I, _I, _P]),
    "cerberus_set_option": (_I, [ctypes.c_char_p, _I]),
    "cerberus_get_option": (_I, [ctypes.c_char_p, ctypes.POINTER(_I)]),
    "cerberus_last_kernel": (ctypes.c_char_p, [_I]),
}


def get():
    """Load (once) and return the ctypes handle; raises if the .so is missing."""
    global _lib
    if _lib is not None:
        return _lib
    with _lock:
        if _lib is not None:
            return _lib
        if not os.path.exists(LIB_PATH):
            raise RuntimeError(
                "cerberusnet_amd: %s is missing -- build the HIP extension first "
                "(python -m cerberusnet_amd.build). There is no CPU/PyTorch "
                "fallback for the cerberus:: ops." % LIB_PATH)
        lib = ctypes.CDLL(LIB_PATH)
        for name, (res, args) in PROTOTYPES.items():
            fn = getattr(lib, name)  # AttributeError if the .so is stale
            fn.restype = res
            fn.argtypes = args
        got = lib.cerberus_abi_version()
        if got != ABI_VERSION:
            raise RuntimeError("libcerberus_hip.so ABI %d != expected %d; rebuild"
                               % (got, ABI_VERSION))
        _lib = lib
    return _lib


def check(code: int, what: str):
    """Map a non-zero return code to RuntimeError (reference: AT_ERROR,
    correlation_cuda.cpp:23,40)."""
    if code != 0:
        msg = get().cerberus_error_string(code).decode()
        raise RuntimeError("%s failed: %s (code %d)" % (what, msg, code))


def set_option(key: str, value: int):
    check(get().cerberus_set_option(key.encode(), int(value)), "cerberus_set_option")


def last_kernel(which: int) -> str:
    return get().cerberus_last_kernel(int(which)).decode()
