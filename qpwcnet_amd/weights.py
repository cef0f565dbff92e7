"""Keras <-> torch weight layouts for the ``build_flower`` graph.

No checkpoint ships with the reference (data/.gitignore:1-4; train.py:361 loads from the
author's /tmp), so this module only defines the conversion a user with trained qpwcnet
weights needs: Keras stores Conv2D kernels as (kh, kw, in, out), SeparableConv2D as a
depthwise kernel (kh, kw, in, 1) + pointwise kernel (1, 1, in, out) + bias, Conv2DTranspose
kernels as (kh, kw, out, in), BatchNormalization as gamma/beta/moving_mean/moving_variance
(qpwcnet/core/non_layers.py:196-254, 390-449; by-name loading in qpwcnet/train/util.py:23-54).
The flat names are the ones ``qpwcnet_amd.synth.make_weights`` uses.
"""
import numpy as np


def _kind(name):
    if name.endswith("depthwise.weight"):
        return "depthwise"
    if name.endswith("conv_up.weight"):
        return "transpose"
    if name.endswith(".weight"):
        return "conv"
    return "vector"


def to_keras_layout(weights):
    """torch layouts -> Keras layouts (same flat names)."""
    out = {}
    for k, v in weights.items():
        v = np.asarray(v)
        kind = _kind(k)
        if kind == "conv":            # (out, in, kh, kw) -> (kh, kw, in, out)
            out[k] = np.ascontiguousarray(v.transpose(2, 3, 1, 0))
        elif kind == "depthwise":     # (C, 1, kh, kw) -> (kh, kw, C, 1)
            out[k] = np.ascontiguousarray(v.transpose(2, 3, 0, 1))
        elif kind == "transpose":     # torch (in, out, kh, kw) -> Keras (kh, kw, out, in)
            out[k] = np.ascontiguousarray(v.transpose(2, 3, 1, 0))
        else:
            out[k] = v.copy()
    return out


def from_keras_layout(weights):
    """Keras layouts -> the torch layouts ``build_flower(weights=...)`` expects."""
    out = {}
    for k, v in weights.items():
        v = np.asarray(v)
        kind = _kind(k)
        if kind == "conv":            # (kh, kw, in, out) -> (out, in, kh, kw)
            out[k] = np.ascontiguousarray(v.transpose(3, 2, 0, 1))
        elif kind == "depthwise":     # (kh, kw, C, 1) -> (C, 1, kh, kw)
            out[k] = np.ascontiguousarray(v.transpose(2, 3, 0, 1))
        elif kind == "transpose":     # (kh, kw, out, in) -> (in, out, kh, kw)
            out[k] = np.ascontiguousarray(v.transpose(3, 2, 0, 1))
        else:
            out[k] = v.copy()
    return out


def save_npz(path, weights):
    np.savez_compressed(path, **{k.replace(".", "__"): np.asarray(v) for k, v in weights.items()})


def load_npz(path):
    with np.load(path) as f:
        return {k.replace("__", "."): f[k] for k in f.files}
