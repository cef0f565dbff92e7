#!/usr/bin/env python3
"""Replay one hipGraph of the forward (B=8, 256x512 fp32) 40 times with model attributes set from the command line
(name=value, python literals), for `rocprofv3 --kernel-trace`:  trace_variant.py "dec_chunks=(2,1,4,4)"
(or e.g. "capture_order=('F0','D0','F1','D1','D2','D3','F2','F3','F4')")."""
import ast
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from qpwcnet_amd import metrics, synth  # noqa: E402
from qpwcnet_amd.pwcnet import GraphedForward, build_flower  # noqa: E402

dev = "cuda:0"
hw, B = (256, 512), 8
weights = synth.make_weights(42, hw)
pairs_np, gt_np = synth.make_frames(B, hw[0], hw[1], seed=1234)
pairs = torch.from_numpy(pairs_np).to(dev)
gt_pyr = metrics.multiscale_ground_truth(torch.from_numpy(gt_np).to(dev), [(hw[0] >> s, hw[1] >> s) for s in (5, 4, 3, 2, 1, 0)])
model = build_flower(True, hw, "channels_last", weights=weights, device=dev)
for a in sys.argv[1:]:
    k, v = a.split("=", 1)
    setattr(model, k, ast.literal_eval(v))
g = GraphedForward(model, pairs, epilogue=lambda fl: metrics.per_level_epe(gt_pyr, fl), warmup=2)
for _ in range(40):
    g.replay()
torch.cuda.synchronize()
