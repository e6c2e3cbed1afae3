"""TEST INFRASTRUCTURE — ctypes wrapper of oracle/c/sg_oracle.c (plain-C restatement of the conv / SDFNet math)."""
import ctypes
import os
import subprocess

import numpy as np

_HERE = os.path.dirname(os.path.abspath(__file__))
_LIB = os.path.join(_HERE, "_build", "libsg_oracle.so")
_lib = None


def build():
    subprocess.run(["make", "-C", os.path.join(_HERE, "c")], check=True, capture_output=True)
    return _LIB


def lib():
    global _lib
    if _lib is None:
        if not os.path.exists(_LIB):
            build()
        _lib = ctypes.CDLL(_LIB)
    return _lib


def _p(a):
    return a.ctypes.data_as(ctypes.c_void_p) if a is not None else None


def _f(a):
    return None if a is None else np.ascontiguousarray(a, dtype=np.float32)


def conv_fwd(x, w, b):
    x, w, b = _f(x), _f(w), _f(b)
    N, Ci, D, H, W = x.shape
    Co = w.shape[0]
    y = np.empty((N, Co, D // 2, H // 2, W // 2), np.float32)
    lib().oracle_conv3d_k4s2p1_fwd(_p(x), _p(w), _p(b), _p(y), N, Ci, Co, D, H, W)
    return y


def conv_dgrad(dy, w, b=None):
    dy, w, b = _f(dy), _f(w), _f(b)
    N, Co, OD, OH, OW = dy.shape
    Ci = w.shape[1]
    dx = np.empty((N, Ci, 2 * OD, 2 * OH, 2 * OW), np.float32)
    lib().oracle_conv3d_k4s2p1_dgrad(_p(dy), _p(w), _p(b), _p(dx), N, Ci, Co, 2 * OD, 2 * OH, 2 * OW)
    return dx


def conv_wgrad(dy, x):
    dy, x = _f(dy), _f(x)
    N, Co = dy.shape[:2]
    Ci, D, H, W = x.shape[1:]
    dw = np.empty((Co, Ci, 4, 4, 4), np.float32)
    lib().oracle_conv3d_k4s2p1_wgrad(_p(dy), _p(x), _p(dw), N, Ci, Co, D, H, W)
    return dw


def sdfnet_fwd(points, latent, params):
    points, latent = _f(points), _f(latent)
    params = [_f(p) for p in params]
    N, L = latent.shape
    arr = (ctypes.c_void_p * 16)(*[p.ctypes.data for p in params])
    out = np.empty(N, np.float32)
    lib().oracle_sdfnet_fwd(_p(points), _p(latent), L, arr, _p(out), ctypes.c_long(N))
    return out


def bn_stats(x):
    x = _f(x)
    N, C = x.shape[:2]
    S = x.size // (N * C)
    mean, var = np.empty(C, np.float64), np.empty(C, np.float64)
    lib().oracle_bn_stats(_p(x), _p(mean), _p(var), N, C, ctypes.c_long(S))
    return mean, var
