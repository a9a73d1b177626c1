"""
ORACLE -- TEST INFRASTRUCTURE ONLY.

ctypes driver for oracle/unet_ref.c: runs the reference U-Net
(machine_learning/unet3d.py:77-105 of the reference) with every operator spelled
out in plain C (float64 accumulation), from a numpy state_dict. Slow (direct
loops); meant for tiny widths / patches to pin oracle.reference_path.
"""

import ctypes
import os
import subprocess

import numpy as np

_HERE = os.path.dirname(os.path.abspath(__file__))
_LIB = os.path.join(_HERE, "_build", "libunet_ref.so")
_lib = None

_f = np.ctypeslib.ndpointer(dtype=np.float32, flags="C_CONTIGUOUS")
_i = ctypes.c_int


def lib():
    """Builds (if needed) and loads the C library."""
    global _lib
    if _lib is None:
        if not os.path.exists(_LIB):
            subprocess.run(["make", "-C", _HERE], check=True)
        h = ctypes.CDLL(_LIB)
        h.conv3d_k3p1.argtypes = [_f, _f, _f, _f, _i, _i, _i, _i, _i, _i]
        h.conv3d_k1.argtypes = [_f, _f, _f, _f, _i, _i, _i, _i, _i, _i]
        h.batchnorm_eval.argtypes = [_f, _f, _f, _f, _f, ctypes.c_double, _i, _i, ctypes.c_size_t]
        h.leaky_relu.argtypes = [_f, ctypes.c_double, ctypes.c_size_t]
        h.sigmoid.argtypes = [_f, ctypes.c_size_t]
        h.maxpool2.argtypes = [_f, _f, _i, _i, _i, _i, _i]
        h.upsample2.argtypes = [_f, _f, _i, _i, _i, _i, _i]
        for fn in (h.conv3d_k3p1, h.conv3d_k1, h.batchnorm_eval, h.leaky_relu, h.sigmoid,
                   h.maxpool2, h.upsample2):
            fn.restype = None
        _lib = h
    return _lib


def _c(a):
    return np.ascontiguousarray(a, dtype=np.float32)


def double_conv(x, sd, prefix):
    """DoubleConv (unet3d.py:142-149), BatchNorm in eval mode."""
    h = lib()
    for conv_idx, bn_idx in ((0, 1), (3, 4)):
        w = _c(sd[f"{prefix}.{conv_idx}.weight"])
        b = _c(sd[f"{prefix}.{conv_idx}.bias"])
        n, cin, d, hh, ww = x.shape
        out = np.empty((n, w.shape[0], d, hh, ww), np.float32)
        h.conv3d_k3p1(_c(x), w, b, out, n, cin, w.shape[0], d, hh, ww)
        h.batchnorm_eval(out, _c(sd[f"{prefix}.{bn_idx}.weight"]), _c(sd[f"{prefix}.{bn_idx}.bias"]),
                         _c(sd[f"{prefix}.{bn_idx}.running_mean"]),
                         _c(sd[f"{prefix}.{bn_idx}.running_var"]), 1e-5, n, w.shape[0], d * hh * ww)
        h.leaky_relu(out, 0.01, out.size)
        x = out
    return x


def down(x, sd, name):
    """Down (unet3d.py:194-196): MaxPool3d(2) then DoubleConv."""
    n, c, d, hh, ww = x.shape
    out = np.empty((n, c, d // 2, hh // 2, ww // 2), np.float32)
    lib().maxpool2(_c(x), out, n, c, d, hh, ww)
    return double_conv(out, sd, f"{name}.maxpool_conv.1.double_conv")


def up(x1, x2, sd, name):
    """Up with trilinear upsampling (unet3d.py:247-253, 280-289); sizes that are
    multiples of 16 need no padding, skip goes first in the concatenation."""
    n, c, d, hh, ww = x1.shape
    out = np.empty((n, c, 2 * d, 2 * hh, 2 * ww), np.float32)
    lib().upsample2(_c(x1), out, n, c, d, hh, ww)
    assert out.shape[2:] == x2.shape[2:]
    return double_conv(np.concatenate([x2, out], axis=1), sd, f"{name}.conv.double_conv")


def unet_forward(x, sd):
    """UNet3D.forward (unet3d.py:93-105) -> logits, numpy float32 NCDHW."""
    x1 = double_conv(_c(x), sd, "inc.double_conv")
    x2 = down(x1, sd, "down1")
    x3 = down(x2, sd, "down2")
    x4 = down(x3, sd, "down3")
    x5 = down(x4, sd, "down4")
    y = up(x5, x4, sd, "up1")
    y = up(y, x3, sd, "up2")
    y = up(y, x2, sd, "up3")
    y = up(y, x1, sd, "up4")
    w = _c(sd["outc.conv.weight"]).reshape(sd["outc.conv.weight"].shape[0], -1)
    n, c, d, hh, ww = y.shape
    out = np.empty((n, w.shape[0], d, hh, ww), np.float32)
    lib().conv3d_k1(_c(y), _c(w), _c(sd["outc.conv.bias"]), out, n, c, w.shape[0], d, hh, ww)
    return out
