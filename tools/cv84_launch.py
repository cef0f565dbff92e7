#!/usr/bin/env python3
"""The bench step's own L4 hot-path launches (8x128x256x32 fp32): the cost volume with 84-float output pixels,
the WarpV2, and the fused WarpV2 + cost volume, a dozen times each -- the program tools/make_traffic.sh
profiles for profiles/traffic.json."""
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from qpwcnet_amd import ops  # noqa: E402

g = torch.Generator(device="cuda").manual_seed(0)
prv = torch.randn(8, 128, 256, 32, device="cuda", generator=g)
nxt = torch.randn(8, 128, 256, 32, device="cuda", generator=g)
flo = torch.randn(8, 128, 256, 2, device="cuda", generator=g) * 4
buf = torch.empty(8, 128, 256, 84, device="cuda")
for _ in range(13):
    ops.cost_volume_into(prv, nxt, buf, 0)
    ops.warp(nxt, flo, "clamp")
    ops.cost_volume_into(prv, nxt, buf, 0, flo=flo)
torch.cuda.synchronize()
