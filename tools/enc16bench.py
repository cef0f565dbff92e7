#!/usr/bin/env python3
"""fp16-storage encoder convolution (qpwc_conv3x3_mish_f16_fwd) at the five encoder levels of config 5 (B = 32 pairs =
64 frames) vs library convolution + bias/Mish pass; hipGraph replay of `iters` launches."""
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
NB = 64
for C, H, W in ((16, 128, 256), (32, 64, 128), (64, 32, 64), (128, 16, 32), (256, 8, 16)):
    x = torch.randn(NB, H, W, C, device=dev, generator=g).half()
    w = (torch.randn(C, C, 3, 3, device=dev, generator=g) / (9 * C) ** 0.5).half().contiguous(memory_format=torch.channels_last)
    b = torch.randn(C, device=dev, generator=g)
    taps = ops.conv3x3_taps(w, torch.float16)
    xn = x.permute(0, 3, 1, 2)

    def own():
        return ops.conv3x3_mish(x, taps, b)

    def lib():
        y = F.conv2d(xn, w, None, padding=1)
        return ops.bias_mish_(y.permute(0, 2, 3, 1), b)

    err = float((own().float() - lib().float()).abs().max())
    t1, t2 = timeit(own, 20), timeit(lib, 20)
    mb = 2.0 * NB * H * W * C * 2 / 1e6
    print("C %3d %3dx%3d x %d frames: own %6.1f us (%4.2f TB/s of its %5.1f MB)   library conv + bias/Mish %6.1f us   max|diff| %.1e"
          % (C, H, W, NB, t1, mb / t1, mb, t2, err), flush=True)
