"""Plain-callable twins of the reference's functor layers
(qpwcnet/core/non_layers.py) -- the classes ``pwcnet.py`` actually instantiates.

Hot path (HIP kernels): ``CostVolume``, ``CostVolumeV2``, ``Warp``, ``WarpV2``
(non_layers.py:51-158).  Surrounding blocks (PyTorch-ROCm): ``Split``,
``Upsample``, ``UpConv``, ``DownConv``, ``OptFlow``, ``Flow``, ``UpFlow``.

Tensors cross every block boundary in the declared data format, like the
reference.  Convolutions run on logical-NCHW views; for ``channels_last`` those
views are torch channels_last memory, i.e. nothing is transposed or copied.
"""
import math

import torch
import torch.nn.functional as F

from . import ops
from .backend import CHANNELS_FIRST, CHANNELS_LAST, get_axis, image_data_format


def _get_axis(data_format):
    return get_axis(data_format)


def parse_image_shape(tensor, data_format=None):
    """qpwcnet/core/non_layers.py:19-30."""
    if data_format is None:
        data_format = image_data_format()
    if data_format == CHANNELS_FIRST:
        n, c, h, w = tensor.shape
    else:
        n, h, w, c = tensor.shape
    return {"n": n, "c": c, "h": h, "w": w}


def lrelu(x):
    return F.leaky_relu(x, 0.1)


class _Functor:
    def __init__(self, *args, data_format=None, **kwargs):
        if args or kwargs:
            raise TypeError("unexpected arguments: {} {}".format(args, sorted(kwargs)))
        self.data_format = image_data_format() if data_format is None else data_format
        self.axis = _get_axis(self.data_format)

    # logical-NCHW view for torch convs, and back
    def _nchw(self, x):
        return x.permute(0, 3, 1, 2) if self.data_format == CHANNELS_LAST else x

    def _fmt(self, x):
        return x.permute(0, 2, 3, 1) if self.data_format == CHANNELS_LAST else x


# --------------------------------------------------------------------------
# hot path
# --------------------------------------------------------------------------
class CostVolume(_Functor):
    """qpwcnet/core/non_layers.py:51-104."""

    def __init__(self, search_range=4, *args, **kwargs):
        super().__init__(*args, **kwargs)
        self.search_range = search_range

    def __call__(self, inputs):
        prv, nxt = inputs
        return ops.cost_volume(prv, nxt, self.search_range, self.data_format, 0.1)


class CostVolumeV2(CostVolume):
    """qpwcnet/core/non_layers.py:107-123 (tfa CorrelationCost + lrelu): same kernel."""


class Warp(_Functor):
    """qpwcnet/core/non_layers.py:126-134 -> tf_warp."""

    def __call__(self, inputs):
        img, flo = inputs
        return ops.warp(img, flo, "tfwarp", self.data_format)


class WarpV2(_Functor):
    """qpwcnet/core/non_layers.py:137-158 -> tfa dense_image_warp(img, -flo[..., ::-1])."""

    def __call__(self, inputs):
        img, flo = inputs
        return ops.warp(img, flo, "clamp", self.data_format)


# --------------------------------------------------------------------------
# surrounding network (PyTorch-ROCm)
# --------------------------------------------------------------------------
def _same_pad(size, k, s):
    """TensorFlow 'SAME' padding (before, after) for one spatial dim."""
    out = -(-size // s)
    total = max((out - 1) * s + k - size, 0)
    return total // 2, total - total // 2


def conv2d_same(x, weight, bias, stride):
    """Keras Conv2D(padding='same') on a logical-NCHW tensor.  For stride 2 on even
    sizes the padding is asymmetric (0 before, 1 after), unlike torch padding=1."""
    kh, kw = weight.shape[2], weight.shape[3]
    pt, pb = _same_pad(x.shape[2], kh, stride)
    pl, pr = _same_pad(x.shape[3], kw, stride)
    if pt == pb and pl == pr:
        return F.conv2d(x, weight, bias, stride=stride, padding=(pt, pl))
    return F.conv2d(F.pad(x, (pl, pr, pt, pb)), weight, bias, stride=stride)


class Split(_Functor):
    """qpwcnet/core/non_layers.py:161-168 (tf.split into `num` equal parts)."""

    def __init__(self, num=2, axis=-1, *args, **kwargs):
        super().__init__(*args, **kwargs)
        self.num = num
        self.split_axis = axis

    def __call__(self, x):
        return torch.chunk(x, self.num, dim=self.split_axis)


class Downsample(_Functor):
    """qpwcnet/core/non_layers.py:171-180: AvgPool2D(2x2, padding='same') -- TensorFlow's SAME
    average pooling leaves the padding out of the divisor (== ceil_mode windows clipped at
    the far edge)."""

    def __call__(self, x):
        y = F.avg_pool2d(self._nchw(x), 2, stride=2, ceil_mode=True, count_include_pad=False)
        return self._fmt(y)


class Upsample(_Functor):
    """qpwcnet/core/non_layers.py:183-193: scale * UpSampling2D(2, 'bilinear')
    (half-pixel centres == align_corners=False)."""

    def __init__(self, scale=1.0, *args, **kwargs):
        kwargs.pop("sacle", None)  # reference typo at non_layers.py:468 leaves scale = 1.0
        kwargs.pop("name", None)
        super().__init__(*args, **kwargs)
        self.scale = scale

    def __call__(self, x):
        if (self.data_format == CHANNELS_LAST and x.is_cuda and
                x.dtype in (torch.float32, torch.float16) and x.dim() == 4 and x.shape[3] == 2):
            return ops.upsample2x_flow(x, self.scale)  # flows: one HIP launch
        y = F.interpolate(self._nchw(x), scale_factor=2, mode="bilinear", align_corners=False)
        return self._fmt(y * self.scale)


class _Weighted(_Functor):
    """Block with parameters: looked up by name in a flat ``{name: tensor}`` dict
    (torch layouts) held by the model, so the same weights drive the CPU oracle."""

    def __init__(self, params, prefix, *args, **kwargs):
        super().__init__(*args, **kwargs)
        self.params = params
        self.prefix = prefix

    def p(self, name):
        return self.params[self.prefix + name]

    def p32(self, name):
        """fp32 copy of a (small) parameter, converted once: the HIP epilogues take fp32
        biases whatever the storage dtype of the activations."""
        key = self.prefix + name + "#f32"
        t = self.params.get(key)
        if t is None:
            t = self.params[key] = self.params[self.prefix + name].float().contiguous()
        return t


def _hip_act_ok(y_nchw, data_format):
    """The fused bias+Mish kernel applies to fp32 GPU tensors that are physically NHWC."""
    return (data_format == CHANNELS_LAST and y_nchw.is_cuda and
            y_nchw.dtype in (torch.float32, torch.float16) and
            y_nchw.shape[1] % 4 == 0 and y_nchw.permute(0, 2, 3, 1).is_contiguous())


def _bias_mish(y_nchw, bias, bias32, data_format):
    """Mish(y + bias) for a bias-free library convolution output; HIP epilogue when possible."""
    if _hip_act_ok(y_nchw, data_format):
        ops.bias_mish_(y_nchw.permute(0, 2, 3, 1), bias32)
        return y_nchw
    return F.mish(y_nchw + bias.view(1, -1, 1, 1))


class UpConv(_Weighted):
    """qpwcnet/core/non_layers.py:196-210: Conv2DTranspose(k=4, s=2, 'same') + Mish
    == ConvTranspose2d(k=4, s=2, padding=1)."""

    def __call__(self, x):
        y = F.conv_transpose2d(self._nchw(x), self.p("conv_up.weight"), None, stride=2, padding=1)
        return self._fmt(_bias_mish(y, self.p("conv_up.bias"), self.p32("conv_up.bias"), self.data_format))

    hip_upconv = True
    # the skip half of the concat by qpwc_copy_pixels_fwd instead of tensor.copy_: same step time on the side stream
    # (tools/skipcopy_ab.py: 1.198 vs 1.195 ms/step); round 3: the own kernel is the default, so that the only
    # library launches left on a forward are the two wide pointwise GEMMs of L0 / L1
    skip_copy_hip = True
    fuse_skip_copy = True   # ... and since round 4 by the transposed convolution's own launch where the skip has F channels
    matmul = "f32"      # "bf16x3": the transposed convolution's products as three-way bf16 splits (csrc/split_bf16.h)
    # ... for the input widths where that moves the STEP (tools/step_time.py, one call, ms/step with every other bf16x3
    # kernel on): decoder on the fp32 instructions 1.0935; all four levels split 1.0941; the three coarse levels only
    # 1.1121 (their faster launches get in the way of the coarse flow levels they run beside, as every other change to
    # the second queue did: DESIGN.md 7.0); the finest level (64 channels in) only 1.0829 -- that one
    x3_channels = (64,)

    def _hip_upconv_ok(self, x, skip):
        w = self.p("conv_up.weight")
        return (self.hip_upconv and self.data_format == CHANNELS_LAST and x.is_cuda and
                x.dtype in (torch.float32, torch.float16) and skip.dtype == x.dtype and
                x.is_contiguous() and x.shape[3] in (64, 128, 256) and w.shape[0] == x.shape[3] and
                w.shape[1] % 16 == 0 and tuple(w.shape[2:]) == (4, 4) and skip.shape[3] % 4 == 0 and
                tuple(skip.shape[1:3]) == (2 * x.shape[1], 2 * x.shape[2]))

    def prefill_skip(self, skip, c_in):
        """The concat([up, skip]) buffer of this level with its SKIP half already copied -- for a caller that has the
        skip (an encoder feature) long before the decoder input: QpwcNet lays these copies under the encoder's later
        levels on the side stream instead of beside the coarse flow levels (round 4).  c_in: channels of the decoder
        input the buffer will be completed from.  None where cat_skip() would not take the HIP path."""
        w = self.p("conv_up.weight")
        if not (self.hip_upconv and self.skip_copy_hip and self.data_format == CHANNELS_LAST and skip.is_cuda and
                skip.dtype in (torch.float32, torch.float16) and c_in in (64, 128, 256) and w.shape[0] == c_in and
                w.shape[1] % 16 == 0 and tuple(w.shape[2:]) == (4, 4) and skip.shape[3] % 4 == 0):
            return None
        buf = torch.empty(skip.shape[:3] + (w.shape[1] + skip.shape[3],), dtype=skip.dtype, device=skip.device)
        half = buf[..., w.shape[1]:]
        if not ops.copy_pixels_ok(skip, half):
            return None
        ops.copy_pixels(skip, half)
        return buf

    def cat_skip(self, x, skip, batch_chunks=1, hip_chunks=1, buf=None):
        """concat([UpConv(x), skip]) on the channel axis (pwcnet.py:186-195).  On the HIP path the
        activation epilogue writes its half straight into the concat buffer.  buf: the buffer from prefill_skip()
        (skip half already in place; ordering it before this call is the caller's business)."""
        if self._hip_upconv_ok(x, skip):
            # own transposed-convolution kernel: bias + Mish fused, written straight into the concat buffer
            key = self.prefix + ("#taps_up" if x.dtype == torch.float32 else "#taps_up_f16")
            t = self.params.get(key)
            if t is None:
                t = self.params[key] = ops.upconv_taps(self.p("conv_up.weight"), x.dtype)
            n_up = t.shape[1]
            if self.matmul == "bf16x3" and x.dtype == torch.float32 and x.shape[3] in self.x3_channels:
                t3 = self.params.get(self.prefix + "#taps_up_x3")
                if t3 is None:
                    t3 = self.params[self.prefix + "#taps_up_x3"] = ops.split_bf16x3(t)
                t = t3
            prefilled = buf is not None
            if prefilled and (tuple(buf.shape) != tuple(skip.shape[:3]) + (n_up + skip.shape[3],) or buf.dtype != x.dtype):
                raise ValueError("prefilled concat buffer {} does not fit this level".format(tuple(buf.shape)))
            if not prefilled:
                buf = torch.empty(skip.shape[:3] + (n_up + skip.shape[3],), dtype=x.dtype, device=x.device)
            nch = int(hip_chunks)
            # round 4: the skip half copied by the transposed convolution's own launches (qpwc_upconv4x4s2_mish_cat_fwd)
            cat = (self.fuse_skip_copy and not prefilled and t.dim() == 3 and ops.upconv_cat_ok(x, t, skip, buf))
            if nch > 1 and x.shape[0] % nch == 0:
                nb = x.shape[0] // nch
                for i in range(nch):
                    sl = slice(i * nb, (i + 1) * nb)
                    if cat:
                        ops.upconv4x4s2_mish_cat_into(x[sl], t, self.p32("conv_up.bias"), skip[sl], buf[sl])
                    else:
                        ops.upconv4x4s2_mish_into(x[sl], t, self.p32("conv_up.bias"), buf[sl])
            elif cat:
                ops.upconv4x4s2_mish_cat_into(x, t, self.p32("conv_up.bias"), skip, buf)
            else:
                ops.upconv4x4s2_mish_into(x, t, self.p32("conv_up.bias"), buf)
            half = buf[..., n_up:]
            if prefilled or cat:
                pass
            elif self.skip_copy_hip and ops.copy_pixels_ok(skip, half):
                ops.copy_pixels(skip, half)     # own strided copy instead of the library's elementwise kernel
            else:
                half.copy_(skip)
            return buf
        # (library path) batch_chunks > 1: the transposed convolution as that many launches over slices of the batch (what
        # pwcnet._forward_two_streams asks for when this runs on the side stream beside the coarse flow levels)
        nsplit = int(batch_chunks)
        if (nsplit > 1 and self.data_format == CHANNELS_LAST and x.is_cuda and x.shape[0] % nsplit == 0 and
                skip.shape[3] % 4 == 0 and x.dtype in (torch.float32, torch.float16)):
            w = self.p("conv_up.weight")
            c1, c2 = w.shape[1], skip.shape[3]
            buf = torch.empty(skip.shape[:3] + (c1 + c2,), dtype=x.dtype, device=x.device)
            nb = x.shape[0] // nsplit
            for i in range(nsplit):
                yh = F.conv_transpose2d(self._nchw(x[i * nb:(i + 1) * nb]), w, None, stride=2, padding=1)
                ops.bias_mish_into(yh.permute(0, 2, 3, 1), self.p32("conv_up.bias"), buf[i * nb:(i + 1) * nb], 0)
            buf[..., c1:] = skip
            return buf
        y = F.conv_transpose2d(self._nchw(x), self.p("conv_up.weight"), None, stride=2, padding=1)
        if _hip_act_ok(y, self.data_format) and skip.shape[3] % 4 == 0:
            c1, c2 = y.shape[1], skip.shape[3]
            buf = torch.empty(skip.shape[:3] + (c1 + c2,), dtype=y.dtype, device=y.device)
            ops.bias_mish_into(y.permute(0, 2, 3, 1), self.p32("conv_up.bias"), buf, 0)
            buf[..., c1:] = skip
            return buf
        up = self._fmt(_bias_mish(y, self.p("conv_up.bias"), self.p32("conv_up.bias"), self.data_format))
        return torch.cat([up, skip], dim=self.axis)


class DownConv(_Weighted):
    """qpwcnet/core/non_layers.py:390-449 with use_normalizer=False (pwcnet.py:146):
    3x3 s2 conv + Mish, 3x3 s1 conv + Mish, 3x3 s1 conv + Mish, TF 'same' padding."""

    def __call__(self, x):
        return self.forward_padded(x)[0]

    def forward_padded(self, x, padded_in=None, want_padded=False, after_a=None):
        """-> (features, padded).  `padded_in`: the input already carrying the (0,1) 'SAME'
        padding of the stride-2 conv (then `x` is ignored by conv_a).  `want_padded`: write
        the last activation into a (B,H+1,W+1,C) buffer with a zero border, so that the NEXT
        DownConv needs no pad copy; `features` is then the interior view of that buffer.
        `after_a`: Mish(conv_a(x) + bias) already computed (channels-last), conv_a is skipped."""
        if after_a is None and padded_in is not None and self._hip_s2_ok(padded_in):
            # second level: stride-2 conv + bias + Mish in one HIP launch on the zero-bordered input
            key = self.prefix + ("#taps_a" if padded_in.dtype == torch.float32 else "#taps_a_f16")
            t = self.params.get(key)
            if t is None:
                t = self.params[key] = ops.conv3x3_taps(self.p("conv_a.weight"), padded_in.dtype)
            if self.matmul == "bf16x3" and self.x3_stride2 and padded_in.dtype == torch.float32 and \
                    padded_in.shape[3] in (32, 64, 128):
                t3 = self.params.get(self.prefix + "#taps_a_x3")
                if t3 is None:
                    t3 = self.params[self.prefix + "#taps_a_x3"] = ops.split_bf16x3(t)
                after_a = ops.conv3x3s2_mish_x3(padded_in, t3, self.p32("conv_a.bias"))
            else:
                after_a = ops.conv3x3s2_mish(padded_in, t, self.p32("conv_a.bias"))
        if after_a is not None:
            y = after_a.permute(0, 3, 1, 2)
        else:
            if padded_in is not None:
                y = F.conv2d(self._nchw(padded_in), self.p("conv_a.weight"), None, stride=2)
            else:
                y = conv2d_same(self._nchw(x), self.p("conv_a.weight"), None, 2)
            y = _bias_mish(y, self.p("conv_a.bias"), self.p32("conv_a.bias"), self.data_format)
        h, w = y.shape[2], y.shape[3]
        pad_ok = want_padded and _same_pad(h, 3, 2) == (0, 1) and _same_pad(w, 3, 2) == (0, 1)
        if self._hip_conv_ok(y):
            # narrow levels: conv + bias + Mish (+ the next level's 'SAME' padding) in one HIP launch each
            pad = 1 if pad_ok else 0
            if self.matmul == "bf16x3" and y.dtype == torch.float32:
                taps = self._taps_x3()
                y1 = ops.conv3x3_mish_x3(y.permute(0, 2, 3, 1), taps[0], self.p32("conv_aa.bias"))
                y2 = ops.conv3x3_mish_x3(y1, taps[1], self.p32("conv_b.bias"), pad, pad)
            else:
                taps = self._taps(y.dtype)
                y1 = ops.conv3x3_mish(y.permute(0, 2, 3, 1), taps[0], self.p32("conv_aa.bias"))
                y2 = ops.conv3x3_mish(y1, taps[1], self.p32("conv_b.bias"), pad, pad)
            if pad_ok:
                return y2[:, :h, :w, :], y2
            return y2, None
        y = conv2d_same(y, self.p("conv_aa.weight"), None, 1)
        y = _bias_mish(y, self.p("conv_aa.bias"), self.p32("conv_aa.bias"), self.data_format)
        y = conv2d_same(y, self.p("conv_b.weight"), None, 1)
        if pad_ok and _hip_act_ok(y, self.data_format):
            padded = ops.bias_mish_pad(y.permute(0, 2, 3, 1), self.p32("conv_b.bias"), 1, 1)
            return padded[:, :h, :w, :], padded
        y = _bias_mish(y, self.p("conv_b.bias"), self.p32("conv_b.bias"), self.data_format)
        return self._fmt(y), None

    # own 3x3 + bias + Mish kernels (qpwc_conv3x3_mish_fwd: 16 / 32 channels with the weights in registers,
    # 64 / 128 / 256 with a wave per 16-output block) instead of library convolution + bias/Mish pass
    hip_conv = True
    # fp32 products of those kernels: "f32" = the fp32 matrix instructions, "bf16x3" = three-way bf16 splits of both
    # operands on the bf16 matrix instructions (csrc/split_bf16.h; error per product about one fp32 rounding at worst, 2^-28 on average)
    matmul = "f32"
    x3_stride2 = True   # bf16x3 also for conv_a of levels 3..5 (tools/step_time.py A/B)

    def _hip_s2_ok(self, padded_in):
        w = self.p("conv_a.weight")
        return (self.hip_conv and self.data_format == CHANNELS_LAST and padded_in.is_cuda and
                padded_in.dtype in (torch.float32, torch.float16) and padded_in.is_contiguous() and
                padded_in.shape[3] in (16, 32, 64, 128) and
                tuple(w.shape) == (2 * padded_in.shape[3], padded_in.shape[3], 3, 3) and
                padded_in.shape[1] % 2 == 1 and padded_in.shape[2] % 2 == 1)

    def _hip_conv_ok(self, y_nchw):
        return (self.hip_conv and y_nchw.dtype in (torch.float32, torch.float16) and
                y_nchw.shape[1] in (16, 32, 64, 128, 256) and _hip_act_ok(y_nchw, self.data_format))

    def first_layer(self, pairs, pairs_format=None):
        """enc.0.conv_a on the raw (B,H,W,6) pair -- (B,6,H,W) for pairs_format 'channels_first' --: split,
        frame stacking, 'SAME' padding, stride-2 conv, bias and Mish in one HIP launch -> (2B, H/2, W/2, 16)
        channels-last, or None when that kernel does not apply."""
        w = self.p("conv_a.weight")
        pf = self.data_format if pairs_format is None else pairs_format
        hh, ww, cc = (pairs.shape[2], pairs.shape[3], pairs.shape[1]) if pf == CHANNELS_FIRST else \
            (pairs.shape[1], pairs.shape[2], pairs.shape[3])
        if not (self.hip_conv and self.data_format == CHANNELS_LAST and pairs.is_cuda and
                pairs.dtype in (torch.float32, torch.float16) and tuple(w.shape) == (16, 3, 3, 3) and
                hh % 2 == 0 and ww % 2 == 0 and cc == 6):
            return None
        key = self.prefix + "#taps_a"
        t = self.params.get(key)
        if t is None:
            t = self.params[key] = ops.first_conv_taps(w)
        return ops.first_conv_mish(pairs.contiguous(), t, self.p32("conv_a.bias"), pf)

    def _taps(self, dtype=torch.float32):
        key = self.prefix + ("#taps" if dtype == torch.float32 else "#taps_f16")
        t = self.params.get(key)
        if t is None:
            t = self.params[key] = (ops.conv3x3_taps(self.p("conv_aa.weight"), dtype),
                                    ops.conv3x3_taps(self.p("conv_b.weight"), dtype))
        return t

    def _taps_x3(self):
        key = self.prefix + "#taps_x3"
        t = self.params.get(key)
        if t is None:
            t = self.params[key] = tuple(ops.split_bf16x3(w) for w in self._taps(torch.float32))
        return t


class OptFlow(_Weighted):
    """qpwcnet/core/non_layers.py:213-273: 4x [SeparableConv2D 3x3 + Mish] ->
    1x1 conv + Mish -> BatchNorm (inference, eps 1e-3) -> 3x3 conv (no bias),
    times sqrt(h^2 + w^2) of the input's spatial size."""

    BN_EPS = 1e-3
    # fp32 pointwise products: "f32" = fp32 matrix instructions; "bf16x3" = three-way bf16 splits on the bf16 matrix
    # instructions (qpwc_sepconv3x3_x3_fwd) for the layers where that kernel is the faster one (tools/sepx3bench.py,
    # B=8, us: 118 -> 128 at L4 134 -> 106, at L3 45.5 -> 36.2; 128 -> 64 at L4 78 -> 66, at L3 24.6 -> 21.6; the 32- /
    # 16-output layers +-0 stay on the fp32 instructions: they are bound by staging and the depthwise stage, not by
    # their products)
    matmul = "f32"
    x3_filters = (128, 64)
    x3_min_pixels = 8 * 64 * 128
    # Fused depthwise+pointwise kernel (qpwc_sepconv3x3_fwd: depthwise result stays in LDS, pointwise
    # on the fp32 matrix cores) instead of dwconv + library GEMM: False / True = never / always,
    # None = per layer (_fuse_layer).  tools/sepbench.py, B=8 (us, fused vs depthwise kernel + library GEMM):
    # L4 253 vs 360 for the four layers, L3 84 vs 130.  On the small levels the fused kernel splits a layer's
    # outputs over 2-8 workgroups per tile (round 2): 128 -> 64 at L0 / L1 / L2 9.3 / 9.5 / 12.4 vs 9.8 / 10.8 /
    # 17.9, first layer at L2 (211 channels, 128 tiles) 23.2 vs 22.9 as one launch instead of two; the wide
    # first layers of L0 / L1 (593 / 339 channels on 8 / 32 tiles: 19 / 11 dependent 32-channel steps) stay
    # split: 32.5 vs 12.6 and 22.2 vs 17.1.
    fused_sepconv = None
    # The last two layers + flow head as ONE launch (qpwc_optflow_tail_fwd) up to this many pixels (B*H*W): on the
    # coarse levels the three launches it replaces are bound by their start-up latency, not by their work
    # (8 x 8 tiles with recomputed halos: 2.25 x the matrix work of the 64 -> 32 layer, irrelevant there).
    tail_max_pixels = 16384

    # the split layers' pointwise half by qpwc_pointwise_bias_fwd instead of the library GEMM: built, parity-green, SLOWER -- 15.4 /
    # 12.3 / 23.1 us against 7.5 / 10.6 / 13.1 us for hipBLASLt at the L0 / L1 / L2 shapes (tools/pwbench.py), step 1.131 vs 1.116 ms:
    # with 16 pixels per workgroup every workgroup streams the whole (F, C) weight matrix from L2.  Off; the two library
    # GEMMs of L0 / L1 stay the only library launches of a step.
    own_pointwise = False
    cost84_for_split_fp16 = False   # see wants_cost84: measured +-0 on config 5 (three interleaved pairs of one call): off
    fuse_upsample = False   # set by QpwcNet on its own blocks: flow head + the x2 upsampling of the flow in one launch
    # ... where the two launches are bound by their start-up, not by their work (the fused one recomputes the tile's rim):
    # config 2 (B=8: L3 65 k, L4 262 k pixels) 1.120 vs 1.124 ms/step with it, config 5 (B=32: L4 1 M pixels) 1.626 vs 1.617
    fuse_upsample_max_pixels = 1 << 18
    def _pw_x3(self, i, first84=False):
        key = (i, bool(first84))
        t = self._pw_x3_cache.get(key)
        if t is None:
            t = self._pw_x3_cache[key] = ops.split_bf16x3(self._pw_pad84 if first84 else self._pw_pad[i])
        return t

    @staticmethod
    def _fuse_layer(c_in, n_tiles, fp16=False):
        # fp16 storage (round 4): the split form's multi-source depthwise kernel moves 2 bytes per lane and load (30 us at
        # config 5's L0 / L1), so the fused kernel wins from 128 tiles on for up to 384 channels -- config 5's L1
        # (342 channels, 128 tiles): step 1.615 -> 1.596 ms; its L0 (32 tiles) +-0, stays split
        if fp16 and c_in <= 384 and n_tiles >= 128:
            return True
        return c_in <= 128 or (c_in <= 256 and n_tiles >= 128) or n_tiles >= 256

    def __init__(self, params, prefix, filters=(128, 64, 32, 16), scale=None, *args, **kwargs):
        super().__init__(params, prefix, *args, **kwargs)
        self.filters = tuple(filters)
        self.scale = scale

    # ---- HIP path: multi-source depthwise kernel (no concat, Mish
    # fused on load), pointwise convs as library GEMMs, fused flow head ---------------
    def _prepare_hip(self):
        if getattr(self, "_hip_ready", False):
            return
        self._pw_t, self._pw_b, self._dw = [], [], []
        for i in range(len(self.filters)):
            pw = self.p("feat.{}.pointwise.weight".format(i))
            self._pw_t.append(pw.reshape(pw.shape[0], pw.shape[1]).t().contiguous())
            self._pw_b.append(self.p("feat.{}.bias".format(i)).contiguous())
            dw = self.p("feat.{}.depthwise.weight".format(i))
            self._dw.append(dw.reshape(dw.shape[0], 9).float().contiguous())  # fp32 in every mode
        self._pw_pad = [ops.pad_pointwise(self.p("feat.{}.pointwise.weight".format(i)))
                        for i in range(len(self.filters))]
        self._pw_pad16 = [ops.pad_pointwise(self.p("feat.{}.pointwise.weight".format(i)), torch.float16)
                          for i in range(len(self.filters))]
        # first layer over [cost (81 + 3 zero pads) | ...]: the 84-channel cost volume keeps its pixels
        # 16-byte aligned, the three pad channels get zero weights
        if self._dw[0].shape[0] > 81:
            pw0 = self.p("feat.0.pointwise.weight")
            pw0 = pw0.reshape(pw0.shape[0], -1)
            z = pw0.new_zeros((pw0.shape[0], 3))
            pw84 = torch.cat([pw0[:, :81], z, pw0[:, 81:]], dim=1)
            self._pw_t84 = pw84.t().contiguous()
            self._pw_pad84 = ops.pad_pointwise(pw84)
            self._pw_pad84_16 = ops.pad_pointwise(pw84, torch.float16)
            self._dw84 = torch.cat([self._dw[0][:81], self._dw[0].new_zeros((3, 9)), self._dw[0][81:]]).contiguous()
        self._pw_b32 = [b.float().contiguous() for b in self._pw_b]
        self._pw_x3_cache = {}
        if len(self.filters) == 4:   # dense (F, C) pointwise matrices of the last two layers for qpwc_optflow_tail_fwd
            self._pw3 = self.p("feat.2.pointwise.weight").reshape(self.filters[2], -1).float().contiguous()
            self._pw4 = self.p("feat.3.pointwise.weight").reshape(self.filters[3], -1).float().contiguous()
        self._head = pack_flow_head(self.p("conv.weight"), self.p("conv.bias"), self.p("norm.gamma"),
                                    self.p("norm.beta"), self.p("norm.mean"), self.p("norm.var"),
                                    self.BN_EPS, self.p("flow.weight"))
        self._hip_ready = True

    def wants_cost84(self, prv, other_channels=None):
        """Flow/UpFlow ask: should the cost volume be produced as 84 channels (81 + 3 zero pads)?  A FUSED first
        layer profits (16-byte loads of all three sources) -- and, for fp16 storage, a SPLIT one whose other sources hold
        multiples of 4 channels (`other_channels`: the coarsest level's [cost | prv | nxt]): its depthwise half then runs
        on the 4-channels-per-lane kernel (round 4).  Elsewhere the dense 81-channel volume avoids the pad-zeroing."""
        if not (self.data_format == CHANNELS_LAST and prv.is_cuda and prv.dtype in (torch.float32, torch.float16) and
                prv.shape[3] % 4 == 0 and self.filters[-1] == 16 and self.fused_sepconv is not False):
            return False
        if self.fused_sepconv is True:
            return True
        if self.cost84_for_split_fp16 and prv.dtype == torch.float16 and other_channels is not None and \
                all(c % 4 == 0 for c in other_channels):
            return True
        B, H, W = prv.shape[:3]
        self._prepare_hip()
        return self._fuse_layer(self._dw[0].shape[0], B * ((H + 7) // 8) * ((W + 15) // 16), fp16=prv.dtype == torch.float16)

    def can_use_hip(self, sources):
        return (self.data_format == CHANNELS_LAST and self.filters[-1] == 16 and
                all(t.is_cuda and t.dtype in (torch.float32, torch.float16) for t in sources))

    def from_sources(self, sources):
        """OptFlow on the virtual concat of `sources` ((B,H,W,Ci) each, channels_last)."""
        if not self.can_use_hip(sources):
            y = self(torch.cat(list(sources), dim=self.axis))
            out_format = getattr(self, "out_format", self.data_format)
            if out_format != self.data_format:   # a channels_first model on channels-last blocks wants (B,2,h,w)
                y = y.permute(0, 3, 1, 2) if out_format == CHANNELS_FIRST else y.permute(0, 2, 3, 1)
                y = y.contiguous()
            return y
        self._prepare_hip()
        B, H, W = sources[0].shape[:3]
        scale = self.scale if self.scale is not None else float(H ** 2 + W ** 2) ** 0.5
        z = None
        z_act = False   # z already carries its layer's Mish (fused layers store it activated)
        fp32 = sources[0].dtype == torch.float32
        n_layers = len(self.filters)
        n_tiles = B * ((H + 7) // 8) * ((W + 15) // 16)
        # an 84-channel first source = the zero-padded cost volume (Flow/UpFlow, wants_cost84)
        padded_cost = sources[0].shape[3] == 84 and sum(t.shape[3] for t in sources) == self._dw[0].shape[0] + 3

        def fuse(i):
            if i >= n_layers:
                return False
            if not fp32 and (self._dw[i].shape[0] % 4 if i else not padded_cost):
                return False    # fp16 (qpwc_sepconv3x3_f16_fwd): sources in 8-byte aligned runs of 4 channels
            if self.fused_sepconv is None:
                return self._fuse_layer(self._dw[i].shape[0], n_tiles, fp16=not fp32)
            return bool(self.fused_sepconv)

        use_tail = (fp32 and self.filters == (128, 64, 32, 16) and B * H * W <= self.tail_max_pixels and
                    self.fused_sepconv is not False)
        for i in range(n_layers):
            if use_tail and i == 2:   # z = the second layer's output
                dw = self._dw
                return ops.optflow_tail(z, dw[2], self._pw3, self._pw_b32[2], dw[3], self._pw4, self._pw_b32[3],
                                        self._head, scale, mish_on_load=not z_act,
                                        out_format=getattr(self, "out_format", CHANNELS_LAST))
            src = sources if i == 0 else [z]
            act_in = i > 0 and not z_act
            first84 = i == 0 and padded_cost
            dw_i = self._dw84 if first84 else self._dw[i]
            if fuse(i):  # depthwise + pointwise + bias in one launch, depthwise result stays on chip
                # store Mish(z) when the next consumer is another fused layer (the flow head and the
                # split depthwise kernel take pre-activation tensors and activate on load)
                act_out = fuse(i + 1) or (use_tail and i == 1)
                if fp32 and self.matmul == "bf16x3" and self._pw_pad[i].shape[0] in self.x3_filters \
                        and B * H * W >= self.x3_min_pixels and ops.sepconv3x3_x3_applies(src):
                    pw_i = self._pw_x3(i, first84)
                elif fp32:
                    pw_i = self._pw_pad84 if first84 else self._pw_pad[i]
                else:
                    pw_i = self._pw_pad84_16 if first84 else self._pw_pad16[i]
                z = ops.sepconv3x3(src, dw_i, pw_i, self._pw_b32[i], mish_on_load=act_in, mish_on_store=act_out)
                z_act = act_out
            else:
                y = ops.dwconv3x3(src, dw_i, mish_on_load=act_in)
                pw_i = self._pw_pad84 if first84 else self._pw_pad[i]
                if self.own_pointwise and fp32 and pw_i.shape[0] in (16, 32, 64, 128, 256) and \
                        pw_i.shape[1] == (y.shape[3] + 31) // 32 * 32:
                    # round 4: the pointwise half on the own matrix-core kernel (the library GEMM was the last library
                    # launch of a step: 12-14 us for a start-up bound 1 k / 4 k-row product)
                    z = ops.pointwise_bias(y, pw_i, self._pw_b32[i])
                else:
                    z = torch.addmm(self._pw_b[i], y.view(B * H * W, -1),
                                    self._pw_t84 if first84 else self._pw_t[i]).view(B, H, W, -1)
                z_act = False
        out_format = getattr(self, "out_format", CHANNELS_LAST)
        if self.fuse_upsample and out_format == CHANNELS_LAST and B * H * W <= self.fuse_upsample_max_pixels:
            # round 4: the Upsample(2.0) that follows every level in QpwcNet's flow chain, by the flow head's own launch; the
            # caller finds it on the returned flow (QpwcNet._up)
            out, up = ops.flow_head_up(z, self._head, scale, 2.0)
            out._qpwc_up2 = up
            return out
        return ops.flow_head(z, self._head, scale, out_format)

    def __call__(self, inputs):
        shape = parse_image_shape(inputs, self.data_format)
        scale = self.scale
        if scale is None:
            scale = float(shape["h"] ** 2 + shape["w"] ** 2) ** 0.5
        x = self._nchw(inputs)
        for i in range(len(self.filters)):
            dw = self.p("feat.{}.depthwise.weight".format(i))
            x = F.conv2d(x, dw, None, stride=1, padding=1, groups=dw.shape[0])
            x = F.conv2d(x, self.p("feat.{}.pointwise.weight".format(i)),
                         self.p("feat.{}.bias".format(i)))
            x = F.mish(x)
        x = F.mish(F.conv2d(x, self.p("conv.weight"), self.p("conv.bias")))
        x = F.batch_norm(x, self.p("norm.mean"), self.p("norm.var"), self.p("norm.gamma"),
                         self.p("norm.beta"), training=False, eps=self.BN_EPS)
        f = F.conv2d(x, self.p("flow.weight"), None, stride=1, padding=1)
        return self._fmt(scale * f)


def pack_flow_head(w1, b1, gamma, beta, mean, var, eps, wf):
    """Parameter vector of qpwc_flow_head_fwd (include/qpwc.h): w1[16][16] | b1 | bn_scale |
    bn_shift | wf[ky][kx][in][out]; BatchNorm folded to scale/shift."""
    w1, b1, gamma, beta, mean, var, wf = (t.float() for t in (w1, b1, gamma, beta, mean, var, wf))
    bn_scale = gamma / torch.sqrt(var + eps)
    bn_shift = beta - mean * bn_scale
    return torch.cat([w1.reshape(16, 16).reshape(-1), b1.reshape(-1), bn_scale.reshape(-1),
                      bn_shift.reshape(-1), wf.permute(2, 3, 1, 0).reshape(-1)]).contiguous()


class Flow(_Weighted):
    """First flow block, qpwcnet/core/non_layers.py:315-338:
    cost = cv(prv, nxt); OptFlow(concat[cost, prv, nxt])."""

    def __init__(self, params, prefix, use_tfa=True, hip_optflow=True, *args, **kwargs):
        super().__init__(params, prefix, *args, **kwargs)
        self.flow = OptFlow(params, prefix + "flow.", data_format=self.data_format)
        cls = CostVolumeV2 if use_tfa else CostVolume
        self.cost_volume = cls(data_format=self.data_format)
        self.hip_optflow = bool(hip_optflow)

    def __call__(self, inputs):
        prv, nxt = inputs
        if self.hip_optflow and self.flow.wants_cost84(prv, other_channels=(prv.shape[-1], nxt.shape[-1])):
            cost = _cost84(prv, nxt, self.cost_volume.search_range)
            return self.flow.from_sources((cost, prv, nxt))
        cost = self.cost_volume((prv, nxt))
        if self.hip_optflow:
            return self.flow.from_sources((cost, prv, nxt))
        feat = torch.cat([cost, prv, nxt], dim=self.axis)
        return self.flow(feat)


def _cost84(prv, nxt, search_range, flo=None):
    """The cost volume as (B,H,W,84): 81 channels + 3 zeros, pixels 16-byte aligned (the layout the fused
    first OptFlow layer reads with 16-byte loads).  With `flo`: of prv against WarpV2(nxt, flo), one launch."""
    cost = torch.empty(prv.shape[:3] + (84,), dtype=prv.dtype, device=prv.device)
    ops.cost_volume_into(prv.contiguous(), nxt.contiguous(), cost, 0, search_range, 0.1, flo=flo)
    return cost


# Largest level (bytes of one feature tensor) at which UpFlow fuses WarpV2 into the cost volume.  tools/kbench.py,
# round 3, us (fused vs cost volume + WarpV2): config 2 (B=8 fp32, 34 MB per level) L2 14.6 vs 19.0, L3 22.5 vs 27.5,
# L4 50.1 vs 52.4; config 5 (B=32 fp16, 67 MB) L1 14.5 vs 16.5, L2 21.0 vs 23.5, L3 40.0 vs 42.7, L4 127 vs 113 (pair);
# config 4 (B=16, 1024x2048 fp32) L1 (134 MB) 146 vs 156, but L2 (268 MB) 322 vs 309, L3 (537 MB) 758 vs 684, L4
# (1.07 GB) 1844 vs 1622: past the size the 256 MB memory-side cache holds, the gather's corner loads run up to 17 %
# slower per pixel than at 256x512 -- although the launch's HBM traffic is its algorithmic bytes x 1.05
# (profiles/r03_pmc_warp_cost_volume_L4_c4.txt) -- and the pair wins.
FUSED_FRONT_END_MAX_BYTES = 192 << 20


def fused_kernel_applies(prv, search_range=4):
    """True where qpwc_warp_cost_volume_fwd runs one of the matrix-core kernels for this shape -- asked of the C side
    itself (qpwc_cost_volume_kernel: the launchers' selection rules without a launch), without the size cut."""
    if not (prv.is_cuda and prv.dtype in (torch.float32, torch.float16) and prv.dim() == 4 and search_range == 4):
        return False
    B, H, W, C = prv.shape
    return ops.cost_volume_kernel(B, H, W, C, prv.dtype, fused=True).startswith("cost_volume_mfma_lds")


def fused_front_end_applies(prv, flo=None, search_range=4):
    """True where UpFlow runs WarpV2 + cost volume as ONE launch (qpwc_warp_cost_volume_fwd on the matrix cores: WarpV2
    gathered in the staging step of the workgroup-shared cost-volume kernel): channels-last fp32 or fp16 storage,
    C % 32 == 0, >= 256 regions of 8 x 8 pixels (the C side's rule for that kernel), and a level small enough for the
    fused launch to beat the pair (FUSED_FRONT_END_MAX_BYTES, measured).  Elsewhere two launches are faster."""
    if not fused_kernel_applies(prv, search_range):
        return False
    B, H, W, C = prv.shape
    if prv.dtype == torch.float16 and C < 64:
        # a single 32-channel step (the finest level): since the fp16 WarpV2 moves 16 bytes per lane (round 3: config 5
        # L4 48.5 -> 35.6 us) the pair wins there, 77.8 + 35.6 = 113 us vs 127 fused; from two steps on fused still
        # wins (L3 40.0 vs 23.0 + 19.7, L2 21.0 vs 12.0 + 11.5, L1 14.5 vs 7.2 + 9.3)
        return False
    return B * H * W * C * (2 if prv.dtype == torch.float16 else 4) <= FUSED_FRONT_END_MAX_BYTES


class UpFlow(_Weighted):
    """Refinement block, qpwcnet/core/non_layers.py:341-387:
    nxt_w = WarpV2(nxt, flo); cost = cv(prv, nxt_w); OptFlow(concat[cost, prv, flo]).

    fused (channels_last): WarpV2 and the cost volume as ONE launch (nxt_w never
    exists in memory) feeding OptFlow.from_sources like the unfused form.  None (default): wherever the
    matrix-core fused kernel applies AND beats the pair (fused_front_end_applies, a measured rule); True: wherever
    it applies; False: never."""

    def __init__(self, params, prefix, use_tfa=True, fused=None, hip_optflow=True, *args, **kwargs):
        super().__init__(params, prefix, *args, **kwargs)
        self._config = {"use_tfa": use_tfa}
        self.hip_optflow = bool(hip_optflow)
        self.flow = OptFlow(params, prefix + "flow.", data_format=self.data_format)
        self.warp = WarpV2(data_format=self.data_format)
        cls = CostVolumeV2 if use_tfa else CostVolume
        self.cost_volume = cls(data_format=self.data_format)
        self.fused = (fused is None or bool(fused)) and self.data_format == CHANNELS_LAST and self.hip_optflow
        self.fused_forced = fused is True

    def fuses(self, prv, flo=None):
        """Does this block run WarpV2 + cost volume as one launch for operands like `prv`?"""
        r = self.cost_volume.search_range
        return self.fused and (fused_front_end_applies(prv, flo, r) or
                               (self.fused_forced and fused_kernel_applies(prv, r)))

    def __call__(self, inputs):
        prv, nxt, flo = inputs
        r = self.cost_volume.search_range
        if self.fuses(prv, flo):
            # coordinates are fp32 whatever the storage dtype (fp16 storage: flow_head_up() wrote them beside the flow)
            flo32 = getattr(flo, "_qpwc_f32", None) if flo.dtype == torch.float16 else None
            if flo32 is None:
                flo32 = flo.to(torch.float32).contiguous()
            if self.flow.wants_cost84(prv):
                cost = _cost84(prv, nxt, r, flo=flo32)
            else:
                cost = ops.warp_cost_volume(prv.contiguous(), nxt.contiguous(), flo32, r, 0.1)
            return self.flow.from_sources((cost, prv, flo))
        nxt_w = self.warp((nxt, flo))
        if self.hip_optflow and self.flow.wants_cost84(prv):
            cost = _cost84(prv, nxt_w, self.cost_volume.search_range)
            return self.flow.from_sources((cost, prv, flo))
        cost = self.cost_volume((prv, nxt_w))
        if self.hip_optflow:
            return self.flow.from_sources((cost, prv, flo))
        feat = torch.cat([cost, prv, flo], dim=self.axis)
        return self.flow(feat)


class FrameInterpolate(_Weighted):
    """qpwcnet/core/non_layers.py:276-312 (Keras twin layers.py:356-402): both inputs warped half
    way along their flows (WarpV2), concat [prv_w, nxt_w, flo_01, flo_10 (, img_u)] ->
    SeparableConv2D(64, 3x3, 'same') + Mish -> Conv2D(3, 1x1).

    On the HIP path (channels_last, CUDA tensors) the two big concat members are never copied:
    the depthwise kernel reads [prv_w | nxt_w | small] as a virtual concat, the pointwise convs
    are library GEMMs and bias+Mish is the HIP epilogue."""

    FILTERS = 64

    def __init__(self, params, prefix, up=False, *args, **kwargs):
        kwargs.pop("name", None)  # the functor drops it too (non_layers.py:292)
        super().__init__(params, prefix, *args, **kwargs)
        self.up = bool(up)
        self._config = {"up": self.up}
        self.warp = WarpV2(data_format=self.data_format)

    def get_config(self):
        return dict(self._config)

    def _hip_ok(self, tensors):
        return (self.data_format == CHANNELS_LAST and
                all(t.is_cuda and t.dtype in (torch.float32, torch.float16) for t in tensors))

    def __call__(self, inputs):
        if self.up:
            prv, nxt, flo_01, flo_10, img_u = inputs
        else:
            prv, nxt, flo_01, flo_10 = inputs
            img_u = None
        nxt_w = self.warp((nxt, 0.5 * flo_01))   # "Applying half-scale flow is valid-ish"
        prv_w = self.warp((prv, 0.5 * flo_10))
        rest = [flo_01, flo_10] + ([img_u] if self.up else [])
        return self.head(prv_w, nxt_w, rest)

    def call_stacked(self, swapped, flows, nb, img_u=None):
        """Same block with both warps in one launch: `swapped` = [nxt; prv] and `flows` =
        [flo_01; flo_10] stacked on the batch axis (2*nb)."""
        w = self.warp((swapped, 0.5 * flows))            # [nxt_w; prv_w]
        rest = [flows[:nb], flows[nb:]] + ([img_u] if self.up else [])
        return self.head(w[nb:], w[:nb], rest)

    def head(self, prv_w, nxt_w, rest):
        """conv2(conv1(concat[prv_w, nxt_w, *rest]))."""
        dw = self.p("conv1.depthwise.weight")
        pw = self.p("conv1.pointwise.weight")
        w2 = self.p("conv2.weight")
        if self._hip_ok([prv_w, nxt_w] + list(rest)):
            B, H, W = prv_w.shape[:3]
            small = torch.cat(list(rest), dim=3)
            if prv_w.shape[3] % 4:   # tiny image-level block: one dense source
                src = [torch.cat([prv_w, nxt_w, small], dim=3)]
            else:
                src = [prv_w, nxt_w, small]
            key = self.prefix + "#gemm"
            mats = self.params.get(key)
            if mats is None:
                mats = self.params[key] = (pw.reshape(pw.shape[0], -1).t().contiguous(),
                                           w2.reshape(w2.shape[0], -1).t().contiguous(),
                                           ops.pad_pointwise(pw))
            dw9 = self.p32("conv1.depthwise.weight").reshape(-1, 9)
            n_tiles = B * ((H + 7) // 8) * ((W + 15) // 16)
            if prv_w.dtype == torch.float32 and OptFlow._fuse_layer(dw9.shape[0], n_tiles):
                # SeparableConv2D + Mish in one launch (depthwise result stays on chip)
                z = ops.sepconv3x3(src, dw9, mats[2], self.p32("conv1.bias"), mish_on_store=True)
            else:
                y = ops.dwconv3x3(src, dw9)
                z = torch.mm(y.view(B * H * W, -1), mats[0]).view(B, H, W, -1)
                ops.bias_mish_(z, self.p32("conv1.bias"))
            return torch.addmm(self.p("conv2.bias"), z.view(B * H * W, -1), mats[1]).view(B, H, W, -1)
        x = self._nchw(torch.cat([prv_w, nxt_w] + list(rest), dim=self.axis))
        x = F.conv2d(x, dw, None, stride=1, padding=1, groups=dw.shape[0])
        x = F.mish(F.conv2d(x, pw, self.p("conv1.bias")))
        return self._fmt(F.conv2d(x, w2, self.p("conv2.bias")))


class Flower(_Weighted):
    """qpwcnet/core/non_layers.py:452-505: the flow stack as one callable, so that
    ``build_interpolator`` can run it twice with shared weights (pwcnet.py:268-278).
    The last upsampler is built with the misspelt keyword ``sacle=2.0`` (:468) and therefore
    keeps scale 1.0 -- reproduced: only the final, upsample-only flow is affected."""

    def __init__(self, params, num_layers, output_multiscale=True, use_tfa=True, hip_optflow=True,
                 *args, **kwargs):
        super().__init__(params, "", *args, **kwargs)
        df = self.data_format
        self.num_layers = num_layers
        self.output_multiscale = output_multiscale
        self.use_tfa = use_tfa
        self.flow = Flow(params, "flow.", use_tfa=use_tfa, hip_optflow=hip_optflow, data_format=df)
        self.upsamples = [Upsample(scale=2.0, data_format=df) for _ in range(num_layers)]
        self.upflows = [UpFlow(params, "upflow.{}.".format(i), use_tfa=use_tfa, hip_optflow=hip_optflow,
                               data_format=df) for i in range(num_layers)]
        self.upsamples.append(Upsample(sacle=2.0, data_format=df))

    def __call__(self, inputs):
        enc_prv, enc_nxt, decs_prv, decs_nxt = inputs
        flo_01 = self.flow((enc_prv, enc_nxt))
        flos = [flo_01]
        for i in range(self.num_layers):
            flo_01_u = self.upsamples[i](flo_01)
            flo_01 = self.upflows[i]((decs_prv[i], decs_nxt[i], flo_01_u))
            flos.append(flo_01)
        flo_01 = self.upsamples[-1](flo_01)
        flos.append(flo_01)
        return flos if self.output_multiscale else [flo_01]


def scale_of(h, w):
    return math.sqrt(h * h + w * w)
