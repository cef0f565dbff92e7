#!/usr/bin/env python3
"""The two narrow encoder convolutions (C = 16 at 128x256, C = 32 at 64x128, 16 frames), a few launches each:
a target for tools/pmc.sh (`tools/pmc.sh enc -- python3 tools/encloop.py`)."""
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from qpwcnet_amd import ops  # noqa: E402

dev = "cuda:0"
g = torch.Generator(device=dev).manual_seed(0)
for C, H, W in ((16, 128, 256), (32, 64, 128)):
    x = torch.randn(16, H, W, C, device=dev, generator=g)
    w = (torch.randn(C, C, 3, 3, device=dev, generator=g) / (9 * C) ** 0.5).contiguous(memory_format=torch.channels_last)
    b = torch.randn(C, device=dev, generator=g)
    taps = ops.conv3x3_taps(w)
    for _ in range(6):
        ops.conv3x3_mish(x, taps, b)
torch.cuda.synchronize()
