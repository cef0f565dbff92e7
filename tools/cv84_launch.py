#!/usr/bin/env python3
"""The bench step's own L4 hot-path launches -- the cost volume with 84-element output pixels, the WarpV2 and the
fused WarpV2 + cost volume, a dozen times each -- at the shape of one BASELINE config: the program
tools/make_traffic.sh profiles for profiles/traffic.json.

    cv84_launch.py [--config 2|4|5] [--level 4]
config 2: 8x128x256x32 fp32 (B=8, 256x512); config 4: 16x512x1024x32 fp32 (B=16, 1024x2048);
config 5: 32x128x256x32 fp16 storage (B=32, 256x512)."""
import argparse
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from qpwcnet_amd import ops  # noqa: E402

ap = argparse.ArgumentParser()
ap.add_argument("--config", type=int, default=2, choices=[2, 4, 5])
ap.add_argument("--level", type=int, default=4)
ap.add_argument("--reps", type=int, default=13)
a = ap.parse_args()
B, H0, W0, dt = {2: (8, 256, 512, torch.float32), 4: (16, 1024, 2048, torch.float32),
                 5: (32, 256, 512, torch.float16)}[a.config]
C = [256, 256, 128, 64, 32][a.level]
H, W = H0 >> (5 - a.level), W0 >> (5 - a.level)
g = torch.Generator(device="cuda").manual_seed(0)
prv = torch.randn(B, H, W, C, device="cuda", generator=g).to(dt)
nxt = torch.randn(B, H, W, C, device="cuda", generator=g).to(dt)
flo = torch.randn(B, H, W, 2, device="cuda", generator=g) * 4
buf = torch.empty(B, H, W, 84, device="cuda", dtype=dt)
for _ in range(a.reps if a.config != 4 else min(a.reps, 5)):
    ops.cost_volume_into(prv, nxt, buf, 0)
    ops.warp(nxt, flo, "clamp")
    ops.cost_volume_into(prv, nxt, buf, 0, flo=flo)
torch.cuda.synchronize()
