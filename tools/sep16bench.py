#!/usr/bin/env python3
"""Fused SeparableConv2D for fp16 storage at config 5 level shapes (B = 32; L4 and L3, the four OptFlow layers each):
hipGraph replay timing and bytes moved per second; A/B another build with QPWC_HIP_LIB."""
import os, sys, torch
_root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, _root); sys.path.insert(0, os.path.join(_root, "tools"))
from qpwcnet_amd import ops
from sepbench import timeit
dev = "cuda:0"
g = torch.Generator(device=dev).manual_seed(0)
B = 32
for l, (H, W, Cf) in ((4, (128, 256, 32)), (3, (64, 128, 64))):
    layers = [((84, Cf, 2), 128), ((128,), 64), ((64,), 32), ((32,), 16)]
    for li, (src_ch, F) in enumerate(layers):
        C = sum(src_ch)
        srcs = [torch.randn(B, H, W, c, device=dev, generator=g).half() for c in src_ch]
        dw = torch.randn(C, 9, device=dev, generator=g)
        pw = (torch.randn(F, C, device=dev, generator=g) / C ** 0.5)
        bias = torch.randn(F, device=dev, generator=g)
        pwp = ops.pad_pointwise(pw, torch.float16)
        t = timeit(lambda: ops.sepconv3x3(srcs, dw, pwp, bias, mish_on_load=False, mish_on_store=li < 3), 10)
        mb = B * H * W * (C + F) * 2 / 1e6
        print(os.environ.get("QPWC_HIP_LIB", "product")[-14:], "L%d layer %d C %3d -> F %3d: %7.1f us  %5.2f TB/s" % (l, li + 1, C, F, t, mb / t), flush=True)
