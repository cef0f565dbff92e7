"""torch-CPU restatement of the reference's ``build_flower`` graph
(qpwcnet/core/pwcnet.py:28-67,134-244; blocks qpwcnet/core/non_layers.py) with
the op-for-op hot path of ``oracle/torch_ref.py``.  Test infrastructure and the
cpu_baseline leg only.  PARITY UNPINNED -- see ``oracle/__init__.py``.

Written independently of ``qpwcnet_amd`` (it shares only the weight dictionary):
channels_last tensors throughout, explicit TF 'SAME' padding.
"""
import numpy as np
import torch
import torch.nn.functional as F

from . import torch_ref

ENC = 5
DEC = 4
BN_EPS = 1e-3  # Keras BatchNormalization default


def _c(x):   # NHWC -> NCHW
    return x.permute(0, 3, 1, 2)


def _l(x):   # NCHW -> NHWC
    return x.permute(0, 2, 3, 1)


def _tf_same_conv(x, w, b, stride):
    """Keras Conv2D(padding='same') on NCHW."""
    k = w.shape[2]
    pads = []
    for size in (x.shape[3], x.shape[2]):  # F.pad order: W first
        out = -(-size // stride)
        total = max((out - 1) * stride + k - size, 0)
        pads += [total // 2, total - total // 2]
    return F.conv2d(F.pad(x, pads), w, b, stride=stride)


def mish(x):
    """qpwcnet/core/mish.py:27-28."""
    return x * torch.tanh(F.softplus(x))


class RefNet:
    """storage="fp16": the same graph with the ROUNDING POINTS of an fp16-storage deployment restated
    on it (BASELINE configs[4]; the reference itself has no fp16 path): weights and inputs rounded to
    fp16, every tensor a layer hands to the next one rounded to fp16 (conv + bias + Mish outputs, the
    depthwise and the pointwise halves of a SeparableConv2D, cost volume, warp, flow, upsampled flow),
    all arithmetic in between in fp32.  It is what an fp16 implementation must agree with up to
    accumulation order and the placement of a rounding before or after an activation -- the derived
    bound of tests/test_gpu_configs.py -- whereas its distance from the fp32 graph is fp16's own noise."""

    def __init__(self, weights, dtype=torch.float32, storage="fp32"):
        self.dtype = dtype
        if storage not in ("fp32", "fp16"):
            raise ValueError("storage must be 'fp32' or 'fp16'")
        self.fp16 = storage == "fp16"
        self.w = {k: torch.as_tensor(np.asarray(v)).to(dtype) for k, v in weights.items()}
        if self.fp16:
            self.w = {k: v.half().to(dtype) for k, v in self.w.items()}

    def q(self, x):
        """A storage point: identity in fp32, round-to-nearest fp16 otherwise."""
        return x.half().to(x.dtype) if self.fp16 else x

    def down_conv(self, i, x):                      # non_layers.py:390-449, no normalizer
        y = _c(x)
        for name, s in (("conv_a", 2), ("conv_aa", 1), ("conv_b", 1)):
            p = "enc.{}.{}".format(i, name)
            y = self.q(mish(_tf_same_conv(y, self.w[p + ".weight"], self.w[p + ".bias"], s)))
        return _l(y)

    def up_conv(self, i, x):                        # non_layers.py:196-210
        p = "dec.{}.conv_up".format(i)
        y = F.conv_transpose2d(_c(x), self.w[p + ".weight"], self.w[p + ".bias"], stride=2, padding=1)
        return _l(self.q(mish(y)))

    def opt_flow(self, prefix, feat):               # non_layers.py:213-273
        h, w = feat.shape[1], feat.shape[2]
        scale = float(h ** 2 + w ** 2) ** 0.5
        x = _c(feat)
        for i in range(4):
            dw = self.w["{}feat.{}.depthwise.weight".format(prefix, i)]
            x = self.q(F.conv2d(x, dw, None, padding=1, groups=dw.shape[0]))
            x = F.conv2d(x, self.w["{}feat.{}.pointwise.weight".format(prefix, i)],
                         self.w["{}feat.{}.bias".format(prefix, i)])
            x = self.q(mish(x))
        x = mish(F.conv2d(x, self.w[prefix + "conv.weight"], self.w[prefix + "conv.bias"]))
        g, b = self.w[prefix + "norm.gamma"], self.w[prefix + "norm.beta"]
        m, v = self.w[prefix + "norm.mean"], self.w[prefix + "norm.var"]
        x = (x - m.view(1, -1, 1, 1)) / torch.sqrt(v.view(1, -1, 1, 1) + BN_EPS) \
            * g.view(1, -1, 1, 1) + b.view(1, -1, 1, 1)
        f = F.conv2d(x, self.w[prefix + "flow.weight"], None, padding=1)
        return _l(self.q(scale * f))

    @staticmethod
    def upsample(x, scale):                         # non_layers.py:183-193
        y = F.interpolate(_c(x), scale_factor=2, mode="bilinear", align_corners=False)
        return _l(scale * y)

    @torch.no_grad()
    def __call__(self, inputs):
        """inputs (B,H,W,6) -> list of the 6 multi-scale flows (pwcnet.py:28-67)."""
        x = self.q(torch.as_tensor(inputs).to(self.dtype))
        img_prv, img_nxt = x[..., :3], x[..., 3:]   # Split(2), pwcnet.py:229
        encs = []
        for img in (img_prv, img_nxt):              # pwcnet.py:134-168 (shared weights)
            f, feats = img, [img]
            for i in range(ENC):
                f = self.down_conv(i, f)
                feats.append(f)
            encs.append(feats)
        decs = []
        for feats in encs:                          # pwcnet.py:171-207
            f, out, k = feats[-1], [], -2
            for i in range(DEC):
                f = torch.cat([self.up_conv(i, f), feats[k]], dim=3)
                k -= 1
                out.append(f)
            decs.append(out)
        prv, nxt = encs[0][-1], encs[1][-1]
        cost = self.q(torch_ref.cost_volume(prv, nxt))      # Flow, non_layers.py:332-338
        flo = self.opt_flow("flow.flow.", torch.cat([cost, prv, nxt], dim=3))
        flos = [flo]
        for i in range(DEC):                        # pwcnet.py:43-57
            flo_u = self.q(self.upsample(flo, 2.0))
            prv, nxt = decs[0][i], decs[1][i]
            nxt_w = self.q(torch_ref.warp_v2(nxt, flo_u))   # UpFlow, non_layers.py:377-385
            cost = self.q(torch_ref.cost_volume(prv, nxt_w))
            flo = self.opt_flow("upflow.{}.flow.".format(i), torch.cat([cost, prv, flo_u], dim=3))
            flos.append(flo)
        flos.append(self.q(self.upsample(flo, 2.0)))        # pwcnet.py:60
        return flos


class RefInterpolator(RefNet):
    """``build_interpolator`` (qpwcnet/core/pwcnet.py:247-281) + ``interpolator`` (:70-131) +
    ``FrameInterpolate`` (non_layers.py:276-312), op for op on torch-CPU."""

    def flower(self, enc_prv, enc_nxt, decs_prv, decs_nxt):      # non_layers.py:470-505
        cost = torch_ref.cost_volume(enc_prv, enc_nxt)
        flo = self.opt_flow("flow.flow.", torch.cat([cost, enc_prv, enc_nxt], dim=3))
        flos = [flo]
        for i in range(DEC):
            flo_u = self.upsample(flo, 2.0)
            prv, nxt = decs_prv[i], decs_nxt[i]
            nxt_w = torch_ref.warp_v2(nxt, flo_u)
            cost = torch_ref.cost_volume(prv, nxt_w)
            flo = self.opt_flow("upflow.{}.flow.".format(i), torch.cat([cost, prv, flo_u], dim=3))
            flos.append(flo)
        flos.append(self.upsample(flo, 1.0))        # `Upsample(sacle=2.0)` keeps scale 1.0 (:468)
        return flos

    def frame_interpolate(self, k, prv, nxt, flo_01, flo_10, img_u=None):   # non_layers.py:294-312
        nxt_w = torch_ref.warp_v2(nxt, 0.5 * flo_01)
        prv_w = torch_ref.warp_v2(prv, 0.5 * flo_10)
        feats = [prv_w, nxt_w, flo_01, flo_10] + ([] if img_u is None else [img_u])
        x = _c(torch.cat(feats, dim=3))
        p = "img.{}.".format(k)
        dw = self.w[p + "conv1.depthwise.weight"]
        x = F.conv2d(x, dw, None, padding=1, groups=dw.shape[0])
        x = mish(F.conv2d(x, self.w[p + "conv1.pointwise.weight"], self.w[p + "conv1.bias"]))
        return _l(F.conv2d(x, self.w[p + "conv2.weight"], self.w[p + "conv2.bias"]))

    @staticmethod
    def downsample(x):                              # AvgPool2D(2,2,'same'), non_layers.py:171-180
        h, w = x.shape[1], x.shape[2]
        y = _c(x)
        ones = torch.ones((1, 1, h, w), dtype=x.dtype)
        pad = (0, w % 2, 0, h % 2)                  # SAME: pad after; padding not counted
        s = F.avg_pool2d(F.pad(y, pad), 2, divisor_override=1)
        n = F.avg_pool2d(F.pad(ones, pad), 2, divisor_override=1)
        return _l(s / n)

    @torch.no_grad()
    def __call__(self, inputs):
        """inputs (B,H,W,6) -> list of the 6 multi-scale images (pwcnet.py:127-128)."""
        x = torch.as_tensor(inputs).to(self.dtype)
        img_prv, img_nxt = x[..., :3], x[..., 3:]
        encs = []
        for img in (img_prv, img_nxt):
            f, feats = img, [img]
            for i in range(ENC):
                f = self.down_conv(i, f)
                feats.append(f)
            encs.append(feats)
        decs = []
        for feats in encs:
            f, out, k = feats[-1], [], -2
            for i in range(DEC):
                f = torch.cat([self.up_conv(i, f), feats[k]], dim=3)
                k -= 1
                out.append(f)
            decs.append(out)
        flows_01 = self.flower(encs[1][-1], encs[0][-1], decs[1], decs[0])   # pwcnet.py:271
        flows_10 = self.flower(encs[0][-1], encs[1][-1], decs[0], decs[1])   # pwcnet.py:277
        imgs_prv, imgs_nxt = [img_prv], [img_nxt]
        for _ in range(DEC + 1):                    # pwcnet.py:87-90
            imgs_prv.append(self.downsample(imgs_prv[-1]))
            imgs_nxt.append(self.downsample(imgs_nxt[-1]))
        img = self.frame_interpolate(0, imgs_prv[-1], imgs_nxt[-1], flows_01[0], flows_10[0])
        imgs = [img]
        for i in range(DEC):                        # pwcnet.py:107-121
            img_u = self.upsample(img, 1.0)
            img = self.frame_interpolate(i + 1, decs[0][i], decs[1][i], flows_01[i + 1], flows_10[i + 1],
                                         img_u)
            imgs.append(img)
        imgs.append(self.upsample(img, 1.0))        # pwcnet.py:124
        return imgs


def multiscale_gt(flow_gt, shapes):
    """Per-level ground truth of FlowMseLoss: bilinear resize to (h,w), times h/H
    (qpwcnet/train/loss.py:56-62)."""
    H = flow_gt.shape[1]
    out = []
    for (h, w) in shapes:
        y = F.interpolate(_c(flow_gt), size=(h, w), mode="bilinear", align_corners=False)
        out.append(_l(y) * (h / H))
    return out
