#!/usr/bin/env python3
"""flow_head / upsample2x_flow / warp at the five level shapes (B=8), hipGraph replay of 20 launches each."""
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from qpwcnet_amd import non_layers, ops, synth  # noqa: E402
from sepbench import timeit  # noqa: E402

dev = "cuda:0"
g = torch.Generator(device=dev).manual_seed(0)
w = synth.make_weights(42, (256, 512))
params = {k: torch.as_tensor(v).to(dev) for k, v in w.items()}
of = non_layers.OptFlow(params, "upflow.3.flow.", data_format="channels_last")
of._prepare_hip()
chans = synth.level_channels()
for l in range(5):
    H, W = 256 >> (5 - l), 512 >> (5 - l)
    z = torch.randn(8, H, W, 16, device=dev, generator=g)
    f = torch.randn(8, H, W, 2, device=dev, generator=g)
    img = torch.randn(8, H, W, chans[l], device=dev, generator=g)
    t_head = timeit(lambda: ops.flow_head(z, of._head, 100.0), 20)
    t_up = timeit(lambda: ops.upsample2x_flow(f, 2.0), 20)
    t_warp = timeit(lambda: ops.warp(img, f, "clamp"), 20)
    print("L%d %3dx%3d  flow_head %6.2f us   upsample2x %6.2f us   warp(C=%d) %6.2f us" % (l, H, W, t_head, t_up, chans[l], t_warp))
