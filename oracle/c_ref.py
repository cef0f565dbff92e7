"""ctypes binding of the C oracle (``oracle/c/qpwc_oracle.c``).

Test infrastructure only -- see ``oracle/__init__.py``.  PARITY UNPINNED.
"""
import ctypes
import os
import subprocess

import numpy as np

_HERE = os.path.dirname(os.path.abspath(__file__))
_SO = os.path.join(_HERE, "c", "libqpwc_oracle.so")
_lib = None


def build(force=False):
    src = os.path.join(_HERE, "c", "qpwc_oracle.c")
    if force or not os.path.exists(_SO) or os.path.getmtime(_SO) < os.path.getmtime(src):
        subprocess.check_call(["make", "-C", _HERE, "-s"])
    return _SO


def lib():
    global _lib
    if _lib is None:
        if not os.path.exists(_SO):
            build()
        L = ctypes.CDLL(_SO)
        fp = ctypes.POINTER(ctypes.c_float)
        L.oracle_cost_volume.argtypes = [fp, fp, fp] + [ctypes.c_int] * 7
        L.oracle_cost_volume.restype = ctypes.c_int
        L.oracle_warp.argtypes = [fp, fp, fp] + [ctypes.c_int] * 4 + [
            ctypes.POINTER(ctypes.c_int64), ctypes.c_int, ctypes.c_int]
        L.oracle_warp.restype = ctypes.c_int
        L.oracle_epe.argtypes = [fp, fp] + [ctypes.c_int] * 3
        L.oracle_epe.restype = ctypes.c_double
        L.oracle_correlation_cost.argtypes = [fp, fp, fp] + [ctypes.c_int] * 12
        L.oracle_correlation_cost.restype = ctypes.c_int
        _lib = L
    return _lib


def _fp(a):
    return a.ctypes.data_as(ctypes.POINTER(ctypes.c_float))


def _dims(shape, data_format):
    if data_format == "channels_first":
        B, C, H, W = shape
        return B, H, W, C, 1
    if data_format == "channels_last":
        B, H, W, C = shape
        return B, H, W, C, 0
    raise ValueError("Unsupported data format : {}".format(data_format))


def cost_volume(prv, nxt, search_range=4, data_format="channels_last", acc_double=True):
    prv = np.ascontiguousarray(prv, dtype=np.float32)
    nxt = np.ascontiguousarray(nxt, dtype=np.float32)
    B, H, W, C, layout = _dims(prv.shape, data_format)
    D = (2 * search_range + 1) ** 2
    shape = (B, H, W, D) if layout == 0 else (B, D, H, W)
    out = np.empty(shape, dtype=np.float32)
    rc = lib().oracle_cost_volume(_fp(prv), _fp(nxt), _fp(out), B, H, W, C,
                                  search_range, layout, int(acc_double))
    if rc != 0:
        raise ValueError("oracle_cost_volume rc={}".format(rc))
    return out


def warp(img, flo, data_format="channels_last", mode="clamp"):
    """mode 'clamp' = WarpV2, 'tfwarp' = Warp (tf_warp).  flo may be broadcastable."""
    img = np.ascontiguousarray(img, dtype=np.float32)
    flo = np.ascontiguousarray(flo, dtype=np.float32)
    B, H, W, C, layout = _dims(img.shape, data_format)
    if layout == 0:
        fb, fh, fw, fc = flo.shape
        st = [s // 4 for s in flo.strides]
        strides = [st[0] if fb > 1 else 0, st[1] if fh > 1 else 0,
                   st[2] if fw > 1 else 0, st[3]]
    else:
        fb, fc, fh, fw = flo.shape
        st = [s // 4 for s in flo.strides]
        strides = [st[0] if fb > 1 else 0, st[2] if fh > 1 else 0,
                   st[3] if fw > 1 else 0, st[1]]
    assert fc == 2
    fs = (ctypes.c_int64 * 4)(*strides)
    out = np.empty_like(img)
    rc = lib().oracle_warp(_fp(img), _fp(flo), _fp(out), B, H, W, C, fs, layout,
                           0 if mode == "clamp" else 1)
    if rc != 0:
        raise ValueError("oracle_warp rc={}".format(rc))
    return out


def epe(a, b):
    a = np.ascontiguousarray(a, dtype=np.float32)
    b = np.ascontiguousarray(b, dtype=np.float32)
    B, H, W, _ = a.shape
    return lib().oracle_epe(_fp(a), _fp(b), B, H, W)


def correlation_cost(input_a, input_b, kernel_size=1, max_displacement=4, stride_1=1, stride_2=1, pad=4,
                     data_format="channels_last", lrelu=False):
    """tfa CorrelationCost, scalar C loop (oracle/tfa_ref.py states the algorithm); with
    ``lrelu`` the whole of CostVolumeV2.call (qpwcnet/core/layers.py:128-132)."""
    import math
    a = np.ascontiguousarray(input_a, dtype=np.float32)
    b = np.ascontiguousarray(input_b, dtype=np.float32)
    if a.shape != b.shape:
        raise ValueError("input_a and input_b must have the same shape")
    B, H, W, C, layout = _dims(a.shape, data_format)
    border = max_displacement + (kernel_size - 1) // 2
    oH = int(math.ceil((H + 2 * pad - 2 * border) / float(stride_1)))
    oW = int(math.ceil((W + 2 * pad - 2 * border) / float(stride_1)))
    ds = 2 * (max_displacement // stride_2) + 1
    shape = (B, oH, oW, ds * ds) if layout == 0 else (B, ds * ds, oH, oW)
    out = np.empty(shape, dtype=np.float32)
    rc = lib().oracle_correlation_cost(_fp(a), _fp(b), _fp(out), B, H, W, C, kernel_size, max_displacement,
                                       stride_1, stride_2, pad, layout, layout, int(lrelu))
    if rc != 0:
        raise ValueError("oracle_correlation_cost rc={}".format(rc))
    return out


def cost_volume_v2(prv, nxt, search_range=4, data_format="channels_last"):
    r = int(search_range)
    return correlation_cost(prv, nxt, 1, r, 1, 1, r, data_format, lrelu=True)
