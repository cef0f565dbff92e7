#!/usr/bin/env python3
"""Encoder 3x3 + bias + Mish at the five encoder levels: fp32 matrix instructions (qpwc_conv3x3_mish_fwd) vs the bf16x3
split form (qpwc_conv3x3_mish_x3_fwd); time per launch (hipGraph replay) and the error of both against float64."""
import os
import sys

import torch
import torch.nn.functional as F

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from qpwcnet_amd import ops  # noqa: E402
from sepbench import timeit  # noqa: E402

dev = "cuda:0"
g = torch.Generator(device=dev).manual_seed(0)
for C, H, W in ((16, 128, 256), (32, 64, 128), (64, 32, 64), (128, 16, 32), (256, 8, 16), (32, 37, 50), (64, 19, 23)):
    x = torch.randn(16, H, W, C, device=dev, generator=g) * 3
    w = (torch.randn(C, C, 3, 3, device=dev, generator=g) / (9 * C) ** 0.5)
    b = torch.randn(C, device=dev, generator=g)
    taps = ops.conv3x3_taps(w)
    taps3 = ops.split_bf16x3(taps)
    ref = F.mish(F.conv2d(x.double().permute(0, 3, 1, 2), w.double(), b.double(), padding=1)).permute(0, 2, 3, 1)

    def f32():
        return ops.conv3x3_mish(x, taps, b, 1, 1)

    def x3():
        return ops.conv3x3_mish_x3(x, taps3, b, 1, 1)

    y32, y3 = f32(), x3()
    e32 = float((y32[:, :H, :W].double() - ref).abs().max())
    e3 = float((y3[:, :H, :W].double() - ref).abs().max())
    border = float(y3[:, H:].abs().max() + y3[:, :, W:].abs().max())
    fl = 2.0 * 16 * H * W * C * C * 9
    t1, t2 = timeit(f32, 20), timeit(x3, 20)
    print("C %3d %3dx%3d: fp32 mfma %6.1f us (%5.1f TF) err %.2e | bf16x3 %6.1f us (%5.1f TF) err %.2e border %.1e"
          % (C, H, W, t1, fl / t1 * 1e-6, e32, t2, fl / t2 * 1e-6, e3, border), flush=True)
