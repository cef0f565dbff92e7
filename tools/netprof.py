#!/usr/bin/env python3
"""Per-op GPU time of one network forward: wraps the torch functional ops the model
uses (conv2d, conv_transpose2d, mish, cat, interpolate, batch_norm, pad) and the HIP
hot-path ops with HIP events, keyed by op + shapes.  Dev tool."""
import argparse
import collections
import os
import sys

import torch
import torch.nn.functional as F

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from qpwcnet_amd import ops, synth  # noqa: E402
from qpwcnet_amd.pwcnet import build_flower  # noqa: E402

REC = []


def wrap(mod, name, keyfn):
    orig = getattr(mod, name)

    def f(*a, **k):
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        r = orig(*a, **k)
        e1.record()
        REC.append((keyfn(*a, **k), e0, e1))
        return r
    setattr(mod, name, f)


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--batch", type=int, default=8)
    ap.add_argument("--iters", type=int, default=5)
    ap.add_argument("--fused", action="store_true")
    ap.add_argument("--bench", action="store_true", help="cudnn.benchmark")
    a = ap.parse_args()
    torch.backends.cudnn.benchmark = a.bench
    sh = lambda t: "x".join(map(str, t.shape))
    wrap(F, "conv2d", lambda x, w, b=None, **k: "conv2d in=%s w=%s s=%s g=%s" % (sh(x), sh(w), k.get("stride", 1), k.get("groups", 1)))
    wrap(F, "conv_transpose2d", lambda x, w, b=None, **k: "convT in=%s w=%s" % (sh(x), sh(w)))
    wrap(F, "mish", lambda x: "mish %s" % sh(x))
    wrap(F, "interpolate", lambda x, **k: "interp %s" % sh(x))
    wrap(F, "batch_norm", lambda x, *r, **k: "bn %s" % sh(x))
    wrap(F, "pad", lambda x, p: "pad %s" % sh(x))
    wrap(torch, "addmm", lambda b, x, w: "addmm %s @ %s" % (sh(x), sh(w)))
    wrap(torch, "cat", lambda ts, dim=0: "cat " + "+".join(sh(t) for t in ts))
    dev = "cuda:0"
    model = build_flower(True, (256, 512), "channels_last", device=dev, fused=a.fused)
    pairs, _ = synth.make_frames(a.batch, 256, 512)
    x = torch.from_numpy(pairs).to(dev)
    with torch.no_grad():
        for _ in range(3):
            model(x)
        torch.cuda.synchronize()
        REC.clear()
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        with ops.kernel_timing() as kt:
            e0.record()
            for _ in range(a.iters):
                model(x)
            e1.record()
        torch.cuda.synchronize()
    acc = collections.OrderedDict()
    for k, a0, a1 in REC:
        n, t = acc.get(k, (0, 0.0))
        acc[k] = (n + 1, t + a0.elapsed_time(a1))
    tot = 0.0
    rows = []
    for k, (n, t) in acc.items():
        rows.append((t / a.iters, n // a.iters, k))
        tot += t / a.iters
    hot = sum(n * t for (n, t) in kt.summary().values()) / a.iters
    print("forward (eager, events) %.3f ms; wrapped torch ops %.3f ms; hot path %.3f ms" % (e0.elapsed_time(e1) / a.iters, tot, hot))
    for t, n, k in sorted(rows, reverse=True)[:60]:
        print("%8.1f us  x%d  %s" % (t * 1e3, n, k))
    bycat = collections.defaultdict(float)
    for t, n, k in rows:
        bycat[k.split()[0]] += t
    print({k: round(v, 3) for k, v in bycat.items()})


if __name__ == "__main__":
    main()
