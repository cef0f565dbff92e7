#!/usr/bin/env python3
"""Timing experiment: the encoder (15 launches) on the 2B = 16 stacked frames of a B = 8 batch as one chain,
against two chains of 4 pairs each on two streams (one hipGraph each way; outputs of the split form are not
re-stacked -- this only asks whether two half-size chains overlap their load / matrix / store phases)."""
import os
import sys
import time

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from qpwcnet_amd import synth  # noqa: E402
from qpwcnet_amd.pwcnet import build_flower  # noqa: E402

dev = "cuda:0"
hw, B = (256, 512), 8
weights = synth.make_weights(42, hw)
pairs_np, _ = synth.make_frames(B, hw[0], hw[1], seed=1234)
pairs = torch.from_numpy(pairs_np).to(dev)
model = build_flower(True, hw, "channels_last", weights=weights, device=dev)
nsplit = int(sys.argv[1]) if len(sys.argv) > 1 else 2
parts = [p.contiguous() for p in pairs.chunk(nsplit, dim=0)]
streams = [torch.cuda.Stream() for _ in range(nsplit)]


def whole():
    return model._encode_stacked(pairs)


def split():
    main = torch.cuda.current_stream()
    outs = []
    for s, p in zip(streams, parts):
        s.wait_stream(main)
        with torch.cuda.stream(s):
            outs.append(model._encode_stacked(p))
    for s in streams:
        main.wait_stream(s)
    return outs


def graph_of(fn):
    with torch.no_grad():
        for _ in range(3):
            fn()
        torch.cuda.synchronize()
        side = torch.cuda.Stream()
        side.wait_stream(torch.cuda.current_stream())
        with torch.cuda.stream(side):
            fn()
        torch.cuda.current_stream().wait_stream(side)
        g = torch.cuda.CUDAGraph()
        with torch.cuda.graph(g, capture_error_mode="thread_local"):
            keep = fn()
    return g, keep


gw, kw = graph_of(whole)
gs, ks = graph_of(split)
res = {"whole": [], "split": []}
for rnd in range(5):
    for name, g in (("whole", gw), ("split", gs)):
        for _ in range(5):
            g.replay()
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        for _ in range(50):
            g.replay()
        torch.cuda.synchronize()
        res[name].append((time.perf_counter() - t0) / 50 * 1e6)
for k, v in res.items():
    print(k, " ".join("%.1f" % t for t in v), "median %.1f us" % sorted(v)[len(v) // 2])
