"""Restatement of tensorflow-addons' ``CorrelationCost`` op -- the arithmetic behind the
reference's ``CostVolumeV2`` (qpwcnet/core/layers.py:112-132, twin non_layers.py:107-123),
which is the cost volume the network instantiates by default (``use_tfa=True``,
qpwcnet/core/pwcnet.py:213; non_layers.py:325-326,358-359).

TEST INFRASTRUCTURE ONLY (see ``oracle/__init__.py``).  PARITY UNPINNED.

Why a second, independent statement: ``oracle/np_ref.cost_volume`` restates the in-tree
pure-TF layer (pad + 81 slices + reduce_mean + concat).  The tfa op lives in a third-party
dependency that is NOT under /root/reference: ``tensorflow_addons`` (version unpinned by the
reference: setup.py:17-20, no lock file), op ``tfa.layers.optical_flow.CorrelationCost``
(call sites layers.py:124-125,130; non_layers.py:115-116,121).  Its published algorithm
(tensorflow_addons/custom_ops/layers/cc/kernels/correlation_cost_op.cc, CPU functor, the
FlowNet-C correlation layer) is restated here from the published source as known to the
author of this file -- there is no network, so it could not be re-fetched or executed:

    kernel_rad        = (kernel_size - 1) // 2
    border            = max_displacement + kernel_rad
    out_h             = ceil((H + 2*pad - 2*border) / stride_1)        (same for w)
    disp_rad          = max_displacement // stride_2
    disp_size         = 2*disp_rad + 1;   out channels = disp_size**2
    for n, h, w:   h1 = (h - pad)*stride_1 + max_displacement + kernel_rad   (w1 alike)
      for tj in [-disp_rad, disp_rad]:            # ROW displacement, outer
        for ti in [-disp_rad, disp_rad]:          # COLUMN displacement, inner
          tc = (tj + disp_rad)*disp_size + (ti + disp_rad)
          h2 = h1 + tj*stride_2;  w2 = w1 + ti*stride_2
          acc = 0
          for j, i in kernel window, skipping taps where (h1+j, w1+i) or (h2+j, w2+i)
                  falls outside the UNPADDED image (== zero padding):
            for c: acc += a[n, h1+j, w1+i, c] * b[n, h2+j, w2+i, c]
          out[n, tc, h, w] = acc / (kernel_size**2 * C)

The op's output is NCHW whatever the input format; the Keras layer transposes it back for
``channels_last``.  ``CostVolumeV2.call`` then applies ``leaky_relu(., 0.1)`` (layers.py:131).

With the reference's arguments (kernel_size=1, max_displacement=4, stride_1=stride_2=1,
pad=4: layers.py:124-125) this is  out[n, (tj+4)*9 + (ti+4), h, w] =
(1/C) sum_c a[n,h,w,c] * b[n,h+tj,w+ti,c], zero outside -- which the reference's own test
claims equal to the in-tree layer (qpwcnet/app/test/test_cvol_equal.py:9-25,
test/test_cost_volume.py:16-24).  tests/test_oracle.py asserts that equality between the two
restatements; the GPU tests compare ``CostVolumeV2`` with THIS function.

Three forms, cross-checked in the tests:
  * ``correlation_cost_loops``  -- the 7-deep scalar loop, literally (tiny inputs only);
  * ``correlation_cost``        -- numpy: loops over (tj, ti, j, i, c) with index arithmetic and
                                   bounds masks (no padding, no slicing of a padded copy),
                                   sequential accumulation over c in the input dtype;
  * ``oracle/c/qpwc_oracle.c:oracle_correlation_cost`` -- the scalar loop in C (fp32).
"""
import math

import numpy as np

CHANNELS_LAST = "channels_last"
CHANNELS_FIRST = "channels_first"


def _geometry(H, W, kernel_size, max_displacement, stride_1, stride_2, pad):
    if kernel_size % 2 != 1:
        raise ValueError("kernel_size must be odd")
    kernel_rad = (kernel_size - 1) // 2
    border = max_displacement + kernel_rad
    out_h = int(math.ceil((H + 2 * pad - 2 * border) / float(stride_1)))
    out_w = int(math.ceil((W + 2 * pad - 2 * border) / float(stride_1)))
    if out_h < 1 or out_w < 1:
        raise ValueError("correlation output would be empty")
    disp_rad = max_displacement // stride_2
    return kernel_rad, out_h, out_w, disp_rad, 2 * disp_rad + 1


def _as_nhwc(x, data_format):
    x = np.asarray(x)
    if data_format == CHANNELS_FIRST:
        return np.transpose(x, (0, 2, 3, 1))
    if data_format == CHANNELS_LAST:
        return x
    raise ValueError("Unsupported data format : {}".format(data_format))


def _from_nchw(out_nchw, data_format):
    # the op emits NCHW; the Keras layer transposes for channels_last
    if data_format == CHANNELS_LAST:
        return np.transpose(out_nchw, (0, 2, 3, 1))
    return out_nchw


def correlation_cost_loops(input_a, input_b, kernel_size=1, max_displacement=4, stride_1=1,
                           stride_2=1, pad=4, data_format=CHANNELS_LAST):
    """The published CPU functor, loop for loop.  Pure Python: tiny inputs only."""
    a = _as_nhwc(input_a, data_format)
    b = _as_nhwc(input_b, data_format)
    if a.shape != b.shape:
        raise ValueError("input_a and input_b must have the same shape")
    N, H, W, C = a.shape
    kr, oH, oW, dr, ds = _geometry(H, W, kernel_size, max_displacement, stride_1, stride_2, pad)
    K = a.dtype.type(kernel_size * kernel_size * C)
    out = np.zeros((N, ds * ds, oH, oW), dtype=a.dtype)
    for n in range(N):
        for h in range(oH):
            h1 = (h - pad) * stride_1 + max_displacement + kr
            for w in range(oW):
                w1 = (w - pad) * stride_1 + max_displacement + kr
                for tj in range(-dr, dr + 1):
                    for ti in range(-dr, dr + 1):
                        tc = (tj + dr) * ds + (ti + dr)
                        h2 = h1 + tj * stride_2
                        w2 = w1 + ti * stride_2
                        acc = a.dtype.type(0)
                        for j in range(-kr, kr + 1):
                            if not (0 <= h1 + j < H and 0 <= h2 + j < H):
                                continue
                            for i in range(-kr, kr + 1):
                                if not (0 <= w1 + i < W and 0 <= w2 + i < W):
                                    continue
                                for c in range(C):
                                    acc = acc + a[n, h1 + j, w1 + i, c] * b[n, h2 + j, w2 + i, c]
                        out[n, tc, h, w] = acc / K
    return _from_nchw(out, data_format)


def correlation_cost(input_a, input_b, kernel_size=1, max_displacement=4, stride_1=1, stride_2=1,
                     pad=4, data_format=CHANNELS_LAST):
    """Same algorithm with the three pixel loops vectorised: explicit coordinates, bounds
    masks, clipped gathers and channel-sequential accumulation in the input dtype."""
    a = _as_nhwc(input_a, data_format)
    b = _as_nhwc(input_b, data_format)
    if a.shape != b.shape:
        raise ValueError("input_a and input_b must have the same shape")
    N, H, W, C = a.shape
    kr, oH, oW, dr, ds = _geometry(H, W, kernel_size, max_displacement, stride_1, stride_2, pad)
    dt = a.dtype
    K = dt.type(kernel_size * kernel_size * C)
    h1 = ((np.arange(oH) - pad) * stride_1 + max_displacement + kr)[:, None]   # (oH,1)
    w1 = ((np.arange(oW) - pad) * stride_1 + max_displacement + kr)[None, :]   # (1,oW)
    out = np.zeros((N, ds * ds, oH, oW), dtype=dt)
    for tj in range(-dr, dr + 1):
        for ti in range(-dr, dr + 1):
            tc = (tj + dr) * ds + (ti + dr)
            h2 = h1 + tj * stride_2
            w2 = w1 + ti * stride_2
            acc = np.zeros((N, oH, oW), dtype=dt)
            for j in range(-kr, kr + 1):
                for i in range(-kr, kr + 1):
                    ya, xa, yb, xb = h1 + j, w1 + i, h2 + j, w2 + i
                    ok = (ya >= 0) & (ya < H) & (yb >= 0) & (yb < H) & \
                         (xa >= 0) & (xa < W) & (xb >= 0) & (xb < W)            # (oH,oW)
                    ya, yb = np.clip(ya, 0, H - 1), np.clip(yb, 0, H - 1)
                    xa, xb = np.clip(xa, 0, W - 1), np.clip(xb, 0, W - 1)
                    ya, xa = np.broadcast_arrays(ya, xa)
                    yb, xb = np.broadcast_arrays(yb, xb)
                    okf = ok.astype(dt)[None]
                    for c in range(C):
                        acc = acc + (a[:, ya, xa, c] * b[:, yb, xb, c]) * okf
            out[:, tc] = acc / K
    return _from_nchw(out, data_format)


def leaky_relu(x, alpha=0.1):
    return np.where(x > 0, x, x * np.asarray(alpha, dtype=x.dtype))


def cost_volume_v2(prv, nxt, search_range=4, data_format=CHANNELS_LAST):
    """``CostVolumeV2.call`` -- qpwcnet/core/layers.py:128-132:
    ``leaky_relu(CorrelationCost(1, r, 1, 1, r, data_format)([prv, nxt]), 0.1)``."""
    r = int(search_range)
    return leaky_relu(correlation_cost(prv, nxt, 1, r, 1, 1, r, data_format), 0.1)
