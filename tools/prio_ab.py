#!/usr/bin/env python3
"""A/B of stream priorities under hipGraph capture: whole forward + EPE, B=8 256x512 fp32, one process.
  default : capture stream and decoder side stream at the default priority (the product)
  hi-main : the capture (flow chain) stream created with the highest priority
  lo-side : the decoder side stream created with the lowest priority the runtime offers"""
import os
import sys
import time

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from qpwcnet_amd import metrics, synth  # noqa: E402
from qpwcnet_amd.pwcnet import build_flower  # noqa: E402

dev = "cuda:0"
hw, B = (256, 512), 8
weights = synth.make_weights(42, hw)
pairs_np, gt_np = synth.make_frames(B, hw[0], hw[1], seed=1234)
pairs = torch.from_numpy(pairs_np).to(dev)
gt_pyr = metrics.multiscale_ground_truth(torch.from_numpy(gt_np).to(dev), [(hw[0] >> s, hw[1] >> s) for s in (5, 4, 3, 2, 1, 0)])
lo, hi = torch.cuda.Stream.priority_range()
print("priority range: lowest %d highest %d" % (lo, hi))


def capture(main_prio, side_prio):
    model = build_flower(True, hw, "channels_last", weights=weights, device=dev)
    if side_prio is not None:
        model._side = torch.cuda.Stream(device=dev, priority=side_prio)

    def run():
        with torch.no_grad():
            out = model(pairs)
            return out, metrics.per_level_epe(gt_pyr, out)
    for _ in range(3):
        run()
    torch.cuda.synchronize()
    cap = torch.cuda.Stream(device=dev, priority=main_prio) if main_prio is not None else torch.cuda.Stream(device=dev)
    cap.wait_stream(torch.cuda.current_stream())
    with torch.cuda.stream(cap):
        run()
    torch.cuda.current_stream().wait_stream(cap)
    g = torch.cuda.CUDAGraph()
    with torch.cuda.graph(g, stream=cap, capture_error_mode="thread_local"):
        keep = run()
    return g, keep, model


variants = [("default", None, None), ("hi-main", hi, None), ("lo-side", None, lo), ("hi-main+lo-side", hi, lo)]
graphs = [(n,) + capture(m, s) for n, m, s in variants]
ref = None
for n, g, keep, _ in graphs:
    g.replay()
    torch.cuda.synchronize()
    if ref is None:
        ref = keep[0][-1].clone()
    else:
        assert torch.equal(ref, keep[0][-1]), n
res = {n: [] for n, *_ in graphs}
for rnd in range(4):
    for n, g, keep, _ in graphs:
        for _ in range(5):
            g.replay()
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        for _ in range(40):
            g.replay()
        torch.cuda.synchronize()
        res[n].append((time.perf_counter() - t0) / 40 * 1e3)
for n in res:
    print("%-18s" % n, " ".join("%.4f" % t for t in res[n]), "median %.4f ms" % sorted(res[n])[len(res[n]) // 2])
