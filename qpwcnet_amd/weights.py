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


# ---- Keras by-name layout (qpwcnet/train/util.py:29-50 loads .h5 files with by_name=True) ---------
# A Keras layer created without a name gets `<snake_case class>` / `<...>_<k>` with k counting the
# instances of that class created so far in the process.  `build_flower` in a fresh process creates
# (pwcnet.py:145-162, 179-206, 39-57; non_layers.py:223-254, 402-425, 200-205):
#   encoder : 5 DownConv x (conv_a, conv_aa, conv_b)                    -> conv2d .. conv2d_14
#   decoder : 4 UpConv                                                  -> conv2d_transpose .. _3
#   Flow    : OptFlow = 4 SeparableConv2D, Conv2D 1x1, BatchNormalization, Conv2D 3x3 (no bias)
#   4 UpFlow: the same again each
# (the WarpV2 / Lambda / CorrelationCost layers carry no weights).  Derived from the creation order in
# the reference's source; it could not be checked against a real checkpoint offline (none ships,
# data/.gitignore:1-4, and TensorFlow is not installed here).
def _auto(base, k):
    return base if k == 0 else "{}_{}".format(base, k)


def keras_variable_names():
    """{flat name: 'layer_name/variable_name:0'} for every parameter of ``build_flower``."""
    names = {}
    conv = sep = bn = 0
    for i in range(5):
        for blk in ("conv_a", "conv_aa", "conv_b"):
            layer = _auto("conv2d", conv)
            conv += 1
            names["enc.{}.{}.weight".format(i, blk)] = layer + "/kernel:0"
            names["enc.{}.{}.bias".format(i, blk)] = layer + "/bias:0"
    for i in range(4):
        layer = _auto("conv2d_transpose", i)
        names["dec.{}.conv_up.weight".format(i)] = layer + "/kernel:0"
        names["dec.{}.conv_up.bias".format(i)] = layer + "/bias:0"
    for prefix in ["flow.flow."] + ["upflow.{}.flow.".format(i) for i in range(4)]:
        for j in range(4):
            layer = _auto("separable_conv2d", sep)
            sep += 1
            names["{}feat.{}.depthwise.weight".format(prefix, j)] = layer + "/depthwise_kernel:0"
            names["{}feat.{}.pointwise.weight".format(prefix, j)] = layer + "/pointwise_kernel:0"
            names["{}feat.{}.bias".format(prefix, j)] = layer + "/bias:0"
        layer = _auto("conv2d", conv)
        conv += 1
        names[prefix + "conv.weight"] = layer + "/kernel:0"
        names[prefix + "conv.bias"] = layer + "/bias:0"
        layer = _auto("batch_normalization", bn)
        bn += 1
        for ours, theirs in (("gamma", "gamma"), ("beta", "beta"), ("mean", "moving_mean"),
                             ("var", "moving_variance")):
            names["{}norm.{}".format(prefix, ours)] = "{}/{}:0".format(layer, theirs)
        layer = _auto("conv2d", conv)
        conv += 1
        names[prefix + "flow.weight"] = layer + "/kernel:0"
    return names


def to_keras_named(weights):
    """torch-layout flat weights -> {'layer/variable:0': array in Keras layout}."""
    names = keras_variable_names()
    k = to_keras_layout(weights)
    return {names[n]: v for n, v in k.items()}


def from_keras_named(named):
    """{'layer/variable:0': Keras-layout array} (what h5py yields for a by-name weight file; a
    'layer/layer/variable:0' path is accepted too) -> the torch-layout dict ``build_flower`` takes.
    Raises KeyError naming the first missing variable."""
    short = {}
    for key, v in named.items():
        parts = key.split("/")
        short["/".join(parts[-2:])] = v
    out = {}
    for ours, theirs in keras_variable_names().items():
        if theirs not in short:
            raise KeyError("missing Keras variable '{}' (for '{}')".format(theirs, ours))
        out[ours] = short[theirs]
    return from_keras_layout(out)


def load_keras_h5(path):
    """Read a Keras by-name weight file (qpwcnet/train/util.py:23-54, train.py:383-389).  Needs h5py,
    which this image does not have: fails loudly instead of guessing."""
    try:
        import h5py
    except ImportError as e:  # pragma: no cover - depends on the installation
        raise RuntimeError("load_keras_h5 needs h5py; convert the file to .npz elsewhere and use "
                           "from_keras_named(dict(np.load(...)))") from e
    named = {}

    def visit(name, obj):
        if isinstance(obj, h5py.Dataset):
            named[name] = np.asarray(obj)
    with h5py.File(path, "r") as f:
        (f["model_weights"] if "model_weights" in f else f).visititems(visit)
    return from_keras_named(named)
