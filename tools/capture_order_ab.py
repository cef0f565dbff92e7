#!/usr/bin/env python3
"""A/B of the launch (capture) order of flow levels (F) and decoder levels (D) in QpwcNet._forward_two_streams:
whole forward + EPE under hipGraph, B=8 256x512 fp32, one process."""
import os
import sys
import time

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
cands = ["F0 D0 D1 D2 D3 F1 F2 F3 F4", "F0 D0 F1 D1 D2 D3 F2 F3 F4", "F0 D0 D1 F1 D2 D3 F2 F3 F4",
         "F0 D0 D1 F1 D2 F2 D3 F3 F4", "D0 F0 D1 D2 D3 F1 F2 F3 F4", "F0 D0 F1 D1 F2 D2 F3 D3 F4",
         "F0 D0 D1 D2 F1 D3 F2 F3 F4"]
if len(sys.argv) > 1 and sys.argv[1] == "main":
    # round 3: "M<i>" = decoder level i on the caller's stream (no cross-queue wait for the flow level that needs it)
    cands = ["F0 D0 D1 D2 D3 F1 F2 F3 F4", "M0 F0 D1 D2 D3 F1 F2 F3 F4", "F0 M0 D1 D2 D3 F1 F2 F3 F4",
             "M0 F0 D1 F1 D2 D3 F2 F3 F4", "M0 F0 M1 F1 D2 D3 F2 F3 F4", "M0 D1 F0 D2 D3 F1 F2 F3 F4"]
graphs = []
ref = None
for c in cands:
    model.capture_order = tuple(c.split())
    g = GraphedForward(model, pairs, epilogue=lambda fl: metrics.per_level_epe(gt_pyr, fl), warmup=2)
    graphs.append(g)
    g.replay()
    torch.cuda.synchronize()
    if ref is None:
        ref = [f.clone() for f in g.outputs]
    else:
        assert all(torch.equal(a, b) for a, b in zip(ref, g.outputs)), c
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
    print(c, " ".join("%.4f" % t for t in res[c]), "median %.4f ms" % sorted(res[c])[len(res[c]) // 2])
