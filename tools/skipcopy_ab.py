#!/usr/bin/env python3
"""Whole-forward A/B of the decoder's skip copy: qpwc_copy_pixels_fwd vs tensor.copy_ (hipGraph).
usage: skipcopy_ab.py [f32|f16] [batch]   (default f32 8 = the headline config; f16 32 = config 5)"""
import os
import sys
import time

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from qpwcnet_amd import metrics, synth  # noqa: E402
from qpwcnet_amd.pwcnet import GraphedForward, build_flower  # noqa: E402

dev = "cuda:0"
dt = torch.float16 if len(sys.argv) > 1 and sys.argv[1] == "f16" else torch.float32
hw, B = (256, 512), int(sys.argv[2]) if len(sys.argv) > 2 else 8
weights = synth.make_weights(42, hw)
pairs_np, gt_np = synth.make_frames(B, hw[0], hw[1], seed=1234)
pairs = torch.from_numpy(pairs_np).to(dev).to(dt)
gt_pyr = metrics.multiscale_ground_truth(torch.from_numpy(gt_np).to(dev), [(hw[0] >> s, hw[1] >> s) for s in (5, 4, 3, 2, 1, 0)])
graphs = {}
for name, flag in (("copy_pixels", True), ("tensor.copy_", False)):
    model = build_flower(True, hw, "channels_last", weights=weights, device=dev, dtype=dt)
    for d in model.dec:
        d.skip_copy_hip = flag
    graphs[name] = GraphedForward(model, pairs, epilogue=lambda fl: metrics.per_level_epe(gt_pyr, fl), warmup=2)
res = {n: [] for n in graphs}
for rnd in range(5):
    for n, g in graphs.items():
        for _ in range(5):
            g.replay()
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        for _ in range(40):
            g.replay()
        torch.cuda.synchronize()
        res[n].append((time.perf_counter() - t0) / 40 * 1e3)
for n in res:
    print("%-14s" % n, " ".join("%.4f" % t for t in res[n]), "median %.4f ms" % sorted(res[n])[len(res[n]) // 2])
