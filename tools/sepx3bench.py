#!/usr/bin/env python3
"""Fused fp32 SeparableConv2D of OptFlow at the level shapes: fp32 matrix instructions (qpwc_sepconv3x3_fwd) vs the
bf16x3 split form (qpwc_sepconv3x3_x3_fwd); time per launch (hipGraph replay) and the error of both against float64.

    python tools/sepx3bench.py [--batch 8] [--levels 3,4]"""
import argparse
import os
import sys

import torch
import torch.nn.functional as F_

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from qpwcnet_amd import ops  # noqa: E402
from sepbench import timeit  # noqa: E402

ap = argparse.ArgumentParser()
ap.add_argument("--batch", type=int, default=8)
ap.add_argument("--levels", default="3,4")
ap.add_argument("--iters", type=int, default=20)
a = ap.parse_args()
dev = "cuda:0"
g = torch.Generator(device=dev).manual_seed(0)
chans = [256, 256, 128, 64, 32]
for l in map(int, a.levels.split(",")):
    H, W, Cf = 256 >> (5 - l), 512 >> (5 - l), chans[l]
    B = a.batch
    layers = [((84, Cf, 2), 128), ((128,), 64), ((64,), 32), ((32,), 16)]
    for li, (src_ch, F) in enumerate(layers):
        C = sum(src_ch)
        srcs = [torch.randn(B, H, W, c, device=dev, generator=g) for c in src_ch]
        dw = torch.randn(C, 9, device=dev, generator=g) / 3
        pw = torch.randn(F, C, device=dev, generator=g) / C ** 0.5
        bias = torch.randn(F, device=dev, generator=g)
        pwp = ops.pad_pointwise(pw)
        pw3 = ops.split_bf16x3(pwp)
        for act in ((False, True), (True, False)) if li == 1 else ((False, li < 3),):
            x = torch.cat(srcs, dim=3).double().permute(0, 3, 1, 2)
            if act[0]:
                x = F_.mish(x)
            y = F_.conv2d(x, dw.double().view(C, 1, 3, 3), None, padding=1, groups=C)
            ref = F_.conv2d(y, pw.double().view(F, C, 1, 1), bias.double()).permute(0, 2, 3, 1)
            if act[1]:
                ref = F_.mish(ref)

            def f32():
                return ops.sepconv3x3(srcs, dw, pwp, bias, mish_on_load=act[0], mish_on_store=act[1])

            def x3():
                return ops.sepconv3x3(srcs, dw, pw3, bias, mish_on_load=act[0], mish_on_store=act[1])

            e32 = float((f32().double() - ref).abs().max())
            e3 = float((x3().double() - ref).abs().max())
            t1, t2 = timeit(f32, a.iters), timeit(x3, a.iters)
            print("L%d %3d -> %3d act %d%d: fp32 mfma %6.1f us err %.2e | bf16x3 %6.1f us err %.2e   (|ref| max %.1f)"
                  % (l, C, F, act[0], act[1], t1, e32, t2, e3, float(ref.abs().max())), flush=True)
