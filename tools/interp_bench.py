#!/usr/bin/env python3
"""Throughput of the frame-interpolation model (qpwcnet build_interpolator,
pwcnet.py:247-281) on one MI355X: hipGraph replay of the whole forward, frames resident in HBM.

    python tools/interp_bench.py [--batch 8] [--steps 50] [--warmup 10]

Prints one JSON line: pairs/s and ms/step.  (Parity of this model against the CPU oracle is
tests/test_gpu_interpolator.py's job; oracle/ is test infrastructure and is not used here.)"""
import argparse
import json
import os
import sys
import time

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from qpwcnet_amd import synth  # noqa: E402
from qpwcnet_amd.pwcnet import GraphedForward, build_interpolator  # noqa: E402


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--batch", type=int, default=8)
    ap.add_argument("--steps", type=int, default=50)
    ap.add_argument("--warmup", type=int, default=10)
    ap.add_argument("--height", type=int, default=256)
    ap.add_argument("--width", type=int, default=512)
    ap.add_argument("--cpu-pairs", type=int, default=0, help="ignored (kept for old command lines)")
    a = ap.parse_args()
    hw = (a.height, a.width)
    dev = torch.device("cuda:0")
    torch.backends.cudnn.benchmark = True
    weights = synth.make_interpolator_weights(42, hw)
    pairs_np, _ = synth.make_frames(a.batch, hw[0], hw[1], seed=1234)
    pairs = torch.from_numpy(pairs_np).to(dev)
    model = build_interpolator(hw, "channels_last", weights=weights, device=dev)
    g = GraphedForward(model, pairs)
    for _ in range(a.warmup):
        g.replay()
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(a.steps):
        g.replay()
    torch.cuda.synchronize()
    dt = (time.perf_counter() - t0) / a.steps
    out = {"metric": "interpolated image-pairs/sec", "value": a.batch / dt, "ms_per_step": dt * 1e3,
           "batch": a.batch, "res": "%dx%d" % hw, "dtype": "f32", "data": "synthetic"}
    print(json.dumps(out))


if __name__ == "__main__":
    main()
