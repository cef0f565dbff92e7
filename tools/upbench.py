#!/usr/bin/env python3
"""The decoder's UpConv (Conv2DTranspose 4x4 stride 2 + bias + Mish, qpwc_upconv4x4s2_mish_fwd) at config 2's four decoder
shapes (B = 16 images = 8 pairs): hipGraph replay timing; A/B another build with QPWC_HIP_LIB.  --f16: config 5's (B = 64)."""
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))
from qpwcnet_amd import ops  # noqa: E402
from sepbench import timeit  # noqa: E402

dev = "cuda:0"
f16 = "--f16" in sys.argv
B = 64 if f16 else 16
dt = torch.float16 if f16 else torch.float32
g = torch.Generator(device=dev).manual_seed(0)
tot = 0.0
for H, W, C, F in ((8, 16, 256, 128), (16, 32, 256, 64), (32, 64, 128, 32), (64, 128, 64, 16)):
    x = torch.randn(B, H, W, C, device=dev, generator=g).to(dt)
    w = torch.randn(C, F, 4, 4, device=dev, generator=g) / (4 * C) ** 0.5
    taps = ops.upconv_taps(w, dt)
    b = torch.randn(F, device=dev, generator=g)
    dst = torch.empty(B, 2 * H, 2 * W, 2 * F, device=dev, dtype=dt)
    t = timeit(lambda: ops.upconv4x4s2_mish_into(x, taps, b, dst), 20)
    fl = 2.0 * B * (2 * H) * (2 * W) * 4 * C * F
    tot += t
    print("%-14s C %3d -> F %3d %3dx%-3d: %6.1f us  %5.1f TF" % (os.environ.get("QPWC_HIP_LIB", "product")[-14:], C, F, H, W, t, fl / t / 1e6), flush=True)
print("sum %.1f us" % tot)
