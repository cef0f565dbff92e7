"""Deterministic synthetic weights and frames (no dataset or checkpoint ships with
the reference: data/.gitignore:1-4, train.py:361).

Weights are numpy arrays in torch layouts under flat names, so the GPU model and
the CPU oracle are driven by the very same numbers.  Initialisers start from
the Keras defaults the reference relies on (glorot_uniform kernels, BatchNorm
gamma=1 beta=0 mean=0 var=1 -- non_layers.py:213-254, 390-449) with a fixed
kernel gain and small biases so that activations stay O(1) (see KERNEL_GAIN).
"""
import math

import numpy as np

ENC_FILTERS = (16, 32, 64, 128, 256)   # pwcnet.py:145
DEC_FILTERS = (128, 64, 32, 16)        # pwcnet.py:179
OPTFLOW_FILTERS = (128, 64, 32, 16)    # non_layers.py:215
SEARCH_RANGE = 4


def level_channels():
    """Feature channels at the five hot-path levels L0..L4 (coarse to fine)."""
    chans = [ENC_FILTERS[-1]]
    for i, f in enumerate(DEC_FILTERS):
        chans.append(f + ENC_FILTERS[-2 - i])  # UpConv output ++ encoder skip (pwcnet.py:186-195)
    return chans  # [256, 256, 128, 64, 32]


# Plain glorot_uniform + zero bias lets the activations of an UNTRAINED 30-conv
# Mish network decay to ~1e-6 (flows ~1e-10): numerically degenerate for a parity
# check.  A fixed gain on every kernel that feeds a Mish, and small random
# biases, keep features O(1) like a trained network's.
KERNEL_GAIN = 1.7
BIAS_RANGE = 0.05


def _glorot(rng, shape, fan_in, fan_out, gain=KERNEL_GAIN):
    limit = gain * math.sqrt(6.0 / (fan_in + fan_out))
    return rng.uniform(-limit, limit, size=shape).astype(np.float32)


def _bias(rng, n):
    return rng.uniform(-BIAS_RANGE, BIAS_RANGE, size=(n,)).astype(np.float32)


def _conv(rng, w, prefix, cout, cin, k):
    w[prefix + ".weight"] = _glorot(rng, (cout, cin, k, k), cin * k * k, cout * k * k)
    w[prefix + ".bias"] = _bias(rng, cout)


def _optflow(rng, w, prefix, cin, level_hw, flow_gain):
    c = cin
    for i, f in enumerate(OPTFLOW_FILTERS):
        w["{}feat.{}.depthwise.weight".format(prefix, i)] = _glorot(rng, (c, 1, 3, 3), c * 9, 9)
        w["{}feat.{}.pointwise.weight".format(prefix, i)] = _glorot(rng, (f, c, 1, 1), c, f)
        w["{}feat.{}.bias".format(prefix, i)] = _bias(rng, f)
        c = f
    _conv(rng, w, prefix + "conv", c, c, 1)
    w[prefix + "norm.gamma"] = np.ones((c,), np.float32)
    w[prefix + "norm.beta"] = np.zeros((c,), np.float32)
    w[prefix + "norm.mean"] = np.zeros((c,), np.float32)
    w[prefix + "norm.var"] = np.ones((c,), np.float32)
    fw = _glorot(rng, (2, c, 3, 3), c * 9, 2 * 9, gain=1.0)
    if flow_gain is not None:
        # OptFlow multiplies its output by sqrt(h^2+w^2) (non_layers.py:261-262); a
        # trained head compensates with small weights.  Keep the synthetic flows at
        # a few pixels so the warp gathers look like real inference, not all-border.
        h, w_ = level_hw
        fw *= np.float32(flow_gain / math.sqrt(h * h + w_ * w_))
    w[prefix + "flow.weight"] = fw


def make_weights(seed=42, input_shape=(256, 512), flow_gain=30.0):
    """All parameters of ``build_flower`` (pwcnet.py:210-244), torch layouts."""
    rng = np.random.default_rng(seed)
    w = {}
    cin = 3
    for i, f in enumerate(ENC_FILTERS):
        for name in ("conv_a", "conv_aa", "conv_b"):
            _conv(rng, w, "enc.{}.{}".format(i, name), f, cin, 3)
            cin = f
    chans = level_channels()
    for i, f in enumerate(DEC_FILTERS):
        cin = chans[i]
        # ConvTranspose2d layout (in, out, kh, kw)
        w["dec.{}.conv_up.weight".format(i)] = _glorot(rng, (cin, f, 4, 4), f * 16, cin * 16)
        w["dec.{}.conv_up.bias".format(i)] = _bias(rng, f)
    d2 = (2 * SEARCH_RANGE + 1) ** 2
    H, W = input_shape
    hw = [(H >> (5 - l), W >> (5 - l)) for l in range(5)]
    _optflow(rng, w, "flow.flow.", d2 + 2 * chans[0], hw[0], flow_gain)
    for i in range(4):
        _optflow(rng, w, "upflow.{}.flow.".format(i), d2 + chans[i + 1] + 2, hw[i + 1], flow_gain)
    return w


INTERP_FILTERS = 64   # FrameInterpolate.conv1, non_layers.py:283-285


def make_interpolator_weights(seed=42, input_shape=(256, 512), flow_gain=30.0):
    """``build_interpolator`` (pwcnet.py:247-281): the ``build_flower`` parameters plus the five
    FrameInterpolate blocks img_0..img_4 (SeparableConv2D 64 + Conv2D 3, non_layers.py:283-293).
    Input channels: img_0 = 3+3+2+2; img_k = 2*C_dec + 2 + 2 + 3 (pwcnet.py:101-121)."""
    w = make_weights(seed, input_shape, flow_gain)
    rng = np.random.default_rng(seed + 1000)
    chans = level_channels()
    cins = [3 + 3 + 2 + 2] + [2 * chans[k] + 2 + 2 + 3 for k in range(1, 5)]
    for k, c in enumerate(cins):
        p = "img.{}.".format(k)
        w[p + "conv1.depthwise.weight"] = _glorot(rng, (c, 1, 3, 3), c * 9, 9)
        w[p + "conv1.pointwise.weight"] = _glorot(rng, (INTERP_FILTERS, c, 1, 1), c, INTERP_FILTERS)
        w[p + "conv1.bias"] = _bias(rng, INTERP_FILTERS)
        w[p + "conv2.weight"] = _glorot(rng, (3, INTERP_FILTERS, 1, 1), INTERP_FILTERS, 3, gain=1.0)
        w[p + "conv2.bias"] = _bias(rng, 3)
    return w


def _bilinear_sample(img, yq, xq):
    """Clamp-to-border bilinear sampling in numpy (data generation only)."""
    H, W = img.shape[:2]
    y0 = np.clip(np.floor(yq), 0, H - 2).astype(np.int64)
    x0 = np.clip(np.floor(xq), 0, W - 2).astype(np.int64)
    ay = np.clip(yq - y0, 0, 1)[..., None]
    ax = np.clip(xq - x0, 0, 1)[..., None]
    top = img[y0, x0] * (1 - ax) + img[y0, x0 + 1] * ax
    bot = img[y0 + 1, x0] * (1 - ax) + img[y0 + 1, x0 + 1] * ax
    return top * (1 - ay) + bot * ay


def make_frames(batch, height=256, width=512, seed=1234, max_flow=8.0):
    """-> (pairs (B,H,W,6) f32 in [-0.5,0.5), flow_gt (B,H,W,2) f32, |flow| <= max_flow).

    Whitening as in the reference (x/255 - 0.5: train.py:56-62, test_infer.py:32-37).
    The second frame is the first one displaced by a smooth low-frequency flow so
    that prv[y,x] ~= nxt[y + f_y, x + f_x] (flow channel 0 = x, 1 = y)."""
    rng = np.random.default_rng(seed)
    yy, xx = np.meshgrid(np.arange(height, dtype=np.float64), np.arange(width, dtype=np.float64),
                         indexing="ij")
    pairs = np.empty((batch, height, width, 6), np.float32)
    flows = np.empty((batch, height, width, 2), np.float32)
    for b in range(batch):
        nxt = rng.random((height + 32, width + 32, 3))
        k = 3  # light box blur so that sub-pixel shifts are meaningful
        acc = np.zeros_like(nxt)
        for dy in range(k):
            for dx in range(k):
                acc += np.roll(nxt, (dy - 1, dx - 1), axis=(0, 1))
        nxt = (acc / (k * k))[16:16 + height, 16:16 + width]
        nxt = (nxt - nxt.min()) / (nxt.max() - nxt.min()) - 0.5
        ph = rng.uniform(0, 2 * np.pi, size=4)
        amp = rng.uniform(0.3, 1.0, size=2) * max_flow / math.sqrt(2.0)
        fx = amp[0] * np.sin(2 * np.pi * yy / height + ph[0]) * np.cos(2 * np.pi * xx / width + ph[1])
        fy = amp[1] * np.cos(2 * np.pi * yy / height + ph[2]) * np.sin(2 * np.pi * xx / width + ph[3])
        prv = _bilinear_sample(nxt, yy + fy, xx + fx)
        pairs[b, ..., :3] = prv
        pairs[b, ..., 3:] = nxt
        flows[b, ..., 0] = fx
        flows[b, ..., 1] = fy
    return pairs, flows
