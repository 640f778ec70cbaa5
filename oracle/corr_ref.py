"""ctypes front-end of oracle/corr_oracle.c (TEST INFRASTRUCTURE ONLY).

Restates /root/reference/nnet_training/correlation_package/correlation_cuda.cpp:3-43
(shape maths, zero-initialised outputs) on top of the C restatement of
correlation_cuda_kernel.cu.  numpy in, numpy out; float32 or float64.
"""
import ctypes
import os
import subprocess

import numpy as np

_HERE = os.path.dirname(os.path.abspath(__file__))
_SO = os.path.join(_HERE, "_build", "libcorr_oracle.so")
_lib = None


def build_corr_oracle(force: bool = False) -> str:
    """Compile corr_oracle.c with gcc (seconds).  Returns the .so path."""
    src = os.path.join(_HERE, "corr_oracle.c")
    stale = (not os.path.exists(_SO)
             or os.path.getmtime(_SO) < os.path.getmtime(src))
    if force or stale:
        subprocess.run(["make", "-s", "-C", _HERE, "all"] +
                       (["-B"] if force else []), check=True)
    return _SO


def _load():
    global _lib
    if _lib is None:
        build_corr_oracle()
        lib = ctypes.CDLL(_SO)
        i = ctypes.c_int
        p = ctypes.c_void_p
        lib.corr_oracle_out_shape.argtypes = [i] * 9 + [ctypes.POINTER(i)] * 3
        lib.corr_oracle_out_shape.restype = i
        for suf in ("f32", "f64"):
            f = getattr(lib, "corr_oracle_forward_" + suf)
            f.argtypes = [p, p, p] + [i] * 9
            f.restype = i
            b = getattr(lib, "corr_oracle_backward_" + suf)
            b.argtypes = [p, p, p, p, p] + [i] * 9
            b.restype = i
        _lib = lib
    return _lib


_ERR = {1: "invalid geometry", 2: "unsupported (stride1 != 1 in backward)",
        3: "out of memory"}


def _check(rc):
    if rc:
        raise RuntimeError("corr_oracle: " + _ERR.get(rc, "error %d" % rc))


def corr_out_shape(B, C, H, W, pad, k, d, s1, s2):
    """(oC, oH, oW) as correlation_cuda.cpp:6-14 computes them."""
    lib = _load()
    oc, oh, ow = ctypes.c_int(), ctypes.c_int(), ctypes.c_int()
    _check(lib.corr_oracle_out_shape(B, C, H, W, pad, k, d, s1, s2,
                                     ctypes.byref(oc), ctypes.byref(oh),
                                     ctypes.byref(ow)))
    return oc.value, oh.value, ow.value


def _suffix(a):
    if a.dtype == np.float32:
        return "f32"
    if a.dtype == np.float64:
        return "f64"
    raise TypeError("corr oracle handles float32/float64, got %s" % a.dtype)


def corr_forward_ref(x1, x2, pad, k, d, s1, s2):
    """Reference-order forward; x1/x2: (B,C,H,W) numpy arrays."""
    lib = _load()
    x1 = np.ascontiguousarray(x1)
    x2 = np.ascontiguousarray(x2, dtype=x1.dtype)
    B, C, H, W = x1.shape
    oC, oH, oW = corr_out_shape(B, C, H, W, pad, k, d, s1, s2)
    out = np.zeros((B, oC, oH, oW), dtype=x1.dtype)
    f = getattr(lib, "corr_oracle_forward_" + _suffix(x1))
    _check(f(x1.ctypes.data, x2.ctypes.data, out.ctypes.data, B, C, H, W, pad,
             k, d, s1, s2))
    return out


def corr_backward_ref(x1, x2, gout, pad, k, d, s1, s2):
    """Reference-order backward; returns (grad_input1, grad_input2)."""
    lib = _load()
    x1 = np.ascontiguousarray(x1)
    x2 = np.ascontiguousarray(x2, dtype=x1.dtype)
    gout = np.ascontiguousarray(gout, dtype=x1.dtype)
    B, C, H, W = x1.shape
    oC, oH, oW = corr_out_shape(B, C, H, W, pad, k, d, s1, s2)
    if gout.shape != (B, oC, oH, oW):
        raise ValueError("gradOutput shape %s != %s" %
                         (gout.shape, (B, oC, oH, oW)))
    g1 = np.zeros_like(x1)
    g2 = np.zeros_like(x1)
    f = getattr(lib, "corr_oracle_backward_" + _suffix(x1))
    _check(f(x1.ctypes.data, x2.ctypes.data, gout.ctypes.data, g1.ctypes.data,
             g2.ctypes.data, B, C, H, W, pad, k, d, s1, s2))
    return g1, g2
