#!/usr/bin/env python3
"""End-to-end A/B of OptFlow's per-layer choice between the fused SeparableConv2D and depthwise kernel +
library GEMM (non_layers.OptFlow.fused_sepconv: None = per layer by size, True = always fused), whole
forward under hipGraph, B=8 256x512 fp32."""
import os
import sys
import time

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from qpwcnet_amd import non_layers, synth  # noqa: E402
from qpwcnet_amd.pwcnet import GraphedForward, build_flower  # noqa: E402

dev = torch.device("cuda:0")
torch.backends.cudnn.benchmark = True
hw = (256, 512)
weights = synth.make_weights(42, hw)
pairs = torch.from_numpy(synth.make_frames(8, hw[0], hw[1], seed=1234)[0]).to(dev)
for mode in (None, True, "c128", "c256"):
    if isinstance(mode, str):
        lim = int(mode[1:])
        non_layers.OptFlow.fused_sepconv = None
        non_layers.OptFlow._fuse_layer = staticmethod(lambda c_in, n_tiles, lim=lim: c_in <= lim or n_tiles >= 256)
    else:
        non_layers.OptFlow.fused_sepconv = mode
    model = build_flower(True, hw, "channels_last", weights=weights, device=dev)
    g = GraphedForward(model, pairs)
    for _ in range(10):
        g.replay()
    torch.cuda.synchronize()
    ts = []
    for _ in range(5):
        t0 = time.perf_counter()
        for _ in range(50):
            g.replay()
        torch.cuda.synchronize()
        ts.append((time.perf_counter() - t0) / 50 * 1e3)
    print("fused_sepconv=%s: %.4f ms/step (median of 5 x 50)" % (mode, sorted(ts)[2]))
