#!/usr/bin/env python3
"""A/B of OptFlow.tail_max_pixels (up to which B*H*W the last two SeparableConv2D + flow head run as the one-launch
tail kernel): whole forward + EPE under hipGraph, B=8 256x512 fp32, one process."""
import os
import sys
import time

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from qpwcnet_amd import metrics, non_layers, synth  # noqa: E402
from qpwcnet_amd.pwcnet import GraphedForward, build_flower  # noqa: E402

dev = "cuda:0"
hw, B = (256, 512), 8
weights = synth.make_weights(42, hw)
pairs_np, gt_np = synth.make_frames(B, hw[0], hw[1], seed=1234)
pairs = torch.from_numpy(pairs_np).to(dev)
gt_pyr = metrics.multiscale_ground_truth(torch.from_numpy(gt_np).to(dev), [(hw[0] >> s, hw[1] >> s) for s in (5, 4, 3, 2, 1, 0)])
model = build_flower(True, hw, "channels_last", weights=weights, device=dev)
cands = [0, 1024, 4096, 16384, 65536, 262144]
graphs = []
for c in cands:
    non_layers.OptFlow.tail_max_pixels = c
    graphs.append(GraphedForward(model, pairs, epilogue=lambda fl: metrics.per_level_epe(gt_pyr, fl), warmup=2))
res = {c: [] for c in cands}
for rnd in range(4):
    for c, g in zip(cands, graphs):
        for _ in range(5):
            g.replay()
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        for _ in range(40):
            g.replay()
        torch.cuda.synchronize()
        res[c].append((time.perf_counter() - t0) / 40 * 1e3)
for c in cands:
    print("tail up to %6d px:" % c, " ".join("%.4f" % t for t in res[c]), "median %.4f ms" % sorted(res[c])[len(res[c]) // 2])
