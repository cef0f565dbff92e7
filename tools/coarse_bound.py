#!/usr/bin/env python3
"""Upper bound of what faster coarse levels can buy: the whole forward + EPE under hipGraph,
B=8 256x512 fp32, with the flow blocks of levels 0..k replaced by a constant (no launches at all on those levels) --
a timing experiment only, the flows are wrong.  If the step does not get shorter by what those levels' launches take,
the decoder chain on the second queue is the critical path beside them."""
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


class Const:
    """stands in for Flow / UpFlow: returns a preallocated flow of the level's shape"""
    def __init__(self, h, w):
        self.out = torch.zeros(B, h, w, 2, device=dev)

    def __call__(self, inputs):
        return self.out


variants = {}
for skip in (-1, 0, 1, 2, 3):   # flow levels 0..skip launch nothing
    model = build_flower(True, hw, "channels_last", weights=weights, device=dev)
    if skip >= 0:
        model.flow = Const(hw[0] >> 5, hw[1] >> 5)
    for lv in range(1, skip + 1):
        model.upflows[lv - 1] = Const(hw[0] >> (5 - lv), hw[1] >> (5 - lv))
    variants[skip] = GraphedForward(model, pairs, epilogue=lambda fl: metrics.per_level_epe(gt_pyr, fl), warmup=2)
res = {k: [] for k in variants}
for rnd in range(4):
    for k, g in variants.items():
        for _ in range(5):
            g.replay()
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        for _ in range(40):
            g.replay()
        torch.cuda.synchronize()
        res[k].append((time.perf_counter() - t0) / 40 * 1e3)
for k in variants:
    label = "all levels" if k < 0 else "levels 0..%d launch nothing" % k
    print("%-28s" % label, " ".join("%.4f" % t for t in res[k]), "median %.4f ms" % sorted(res[k])[len(res[k]) // 2])
