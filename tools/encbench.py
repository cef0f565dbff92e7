#!/usr/bin/env python3
"""Encoder 3x3 + bias + Mish kernel (qpwc_conv3x3_mish_fwd) at the two narrow encoder levels vs the
library convolution + bias/Mish pass; hipGraph replay of `iters` launches."""
import os
import sys

import torch
import torch.nn.functional as F

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from qpwcnet_amd import ops  # noqa: E402
from sepbench import timeit  # noqa: E402

torch.backends.cudnn.benchmark = True
dev = "cuda:0"
g = torch.Generator(device=dev).manual_seed(0)
for C, H, W in ((16, 128, 256), (32, 64, 128), (64, 32, 64), (128, 16, 32), (256, 8, 16)):
    x = torch.randn(16, H, W, C, device=dev, generator=g)
    w = (torch.randn(C, C, 3, 3, device=dev, generator=g) / (9 * C) ** 0.5).contiguous(memory_format=torch.channels_last)
    b = torch.randn(C, device=dev, generator=g)
    taps = ops.conv3x3_taps(w)
    xn = x.permute(0, 3, 1, 2)

    def own():
        return ops.conv3x3_mish(x, taps, b)

    def lib():
        y = F.conv2d(xn, w, None, padding=1)
        return ops.bias_mish_(y.permute(0, 2, 3, 1), b)

    err = float((own() - lib()).abs().max())
    fl = 2.0 * 16 * H * W * C * C * 9
    t1, t2 = timeit(own, 20), timeit(lib, 20)
    print("C %2d %dx%d: own %6.1f us (%5.1f TF)   library conv + bias/Mish %6.1f us   max|diff| %.1e"
          % (C, H, W, t1, fl / t1 * 1e-6, t2, err), flush=True)


# stride-2 conv_a of levels 2..5 (C_in -> 2 C_in) on the zero-bordered input vs library convolution + bias/Mish
for CI, H, W in ((16, 128, 256), (32, 64, 128), (64, 32, 64), (128, 16, 32)):
    xp = torch.zeros(16, H + 1, W + 1, CI, device=dev)
    xp[:, :H, :W] = torch.randn(16, H, W, CI, device=dev, generator=g)
    w = (torch.randn(2 * CI, CI, 3, 3, device=dev, generator=g) / (9 * CI) ** 0.5).contiguous(memory_format=torch.channels_last)
    b = torch.randn(2 * CI, device=dev, generator=g)
    taps = ops.conv3x3_taps(w)
    xn = xp.permute(0, 3, 1, 2)

    def own2():
        return ops.conv3x3s2_mish(xp, taps, b)

    def lib2():
        y = F.conv2d(xn, w, None, stride=2)
        return ops.bias_mish_(y.permute(0, 2, 3, 1), b)

    err = float((own2() - lib2()).abs().max())
    t_own, t_lib = timeit(own2, 20), timeit(lib2, 20)
    fl = 16 * (H // 2) * (W // 2) * 2 * CI * 9 * CI * 2
    print("s2 C %3d -> %3d %3dx%3d: own %6.1f us (%5.1f TF)   library conv + bias/Mish %6.1f us   max|diff| %.1e" % (
        CI, 2 * CI, H, W, t_own, fl / t_own / 1e6, t_lib, err))
