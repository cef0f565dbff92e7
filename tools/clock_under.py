#!/usr/bin/env python3
"""Shader clock the chip sustains under one kernel at a time (qpwc_clock_probe beside a hipGraph of back-to-back
launches): is a matrix-bound kernel at 0.5 of the 2.4 GHz peak, or at 0.7 of what the clock it is given can deliver?"""
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from qpwcnet_amd import _hip, ops  # noqa: E402

dev = "cuda:0"
g = torch.Generator(device=dev).manual_seed(0)


def rnd(*s):
    return torch.randn(*s, device=dev, generator=g)


def clock_under(name, fn, flops=None, n_launch=40):
    for _ in range(3):
        fn()
    torch.cuda.synchronize()
    side = torch.cuda.Stream()
    gr = torch.cuda.CUDAGraph()
    with torch.cuda.stream(side):
        with torch.cuda.graph(gr, stream=side, capture_error_mode="thread_local"):
            for _ in range(n_launch):
                fn()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    for _ in range(20):
        gr.replay()
    e0.record()
    for _ in range(10):
        gr.replay()
    e1.record()
    e1.synchronize()
    us = e0.elapsed_time(e1) / (10 * n_launch) * 1e3
    reps = max(20, int(80e3 / (us * n_launch)))      # ~80 ms of launches
    n = 2000
    buf = torch.zeros(2 * n, dtype=torch.int64, device=dev)
    probe = torch.cuda.Stream()
    for _ in range(reps // 2):
        gr.replay()
    probe.wait_stream(torch.cuda.current_stream())
    with torch.cuda.stream(probe):
        _hip.check(_hip.lib().qpwc_clock_probe(buf.data_ptr(), n, 4, probe.cuda_stream))
    for _ in range(reps):
        gr.replay()
    torch.cuda.synchronize()
    v = buf.cpu().view(n, 2).double()
    dt, dr = v[1:, 0] - v[:-1, 0], v[1:, 1] - v[:-1, 1]
    mhz = (dt / dr * 100.0).sort().values
    med = float(mhz[n // 2])
    line = "%-44s %8.1f us   clock median %6.0f MHz (p10 %6.0f, p90 %6.0f)" % (name, us, med, float(mhz[n // 10]), float(mhz[9 * n // 10]))
    if flops:
        tf = flops / us * 1e-6
        peak = 256 * 4 * 64 * med * 1e6 / 1e12
        line += "   %6.1f TF = %.2f of the 157.3 TF peak, %.2f of the %.0f TF the pipe delivers at that clock" % (tf, tf / 157.3, tf / peak, peak)
    print(line, flush=True)


def idle():
    return None


x = rnd(16)
clock_under("tiny elementwise (idle chip)", lambda: x.add_(1.0))
for C, H, W in ((16, 128, 256), (64, 32, 64), (256, 8, 16)):
    xx = rnd(16, H, W, C)
    taps = ops.conv3x3_taps(rnd(C, C, 3, 3) / (9 * C) ** 0.5)
    b = rnd(C)
    clock_under("encoder conv3x3 C=%d" % C, lambda: ops.conv3x3_mish(xx, taps, b), 2.0 * 16 * H * W * C * C * 9)
B, H, W = 8, 128, 256
for chans, F in (((84, 32, 2), 128), ((128,), 64)):
    C = sum(chans)
    srcs = [rnd(B, H, W, c) for c in chans]
    dw, pw, bias = rnd(C, 9), rnd(F, C) / C ** 0.5, rnd(F)
    pwp = ops.pad_pointwise(pw)
    clock_under("sepconv3x3 L4 %d -> %d" % (C, F), lambda: ops.sepconv3x3(srcs, dw, pwp, bias, mish_on_store=True),
                B * H * W * C * (2.0 * F + 18))
prv, nxt, flo = rnd(B, H, W, 32), rnd(B, H, W, 32), rnd(B, H, W, 2) * 3
clock_under("cost volume L4", lambda: ops.cost_volume(prv, nxt), B * H * W * 81 * 2.0 * 32)
clock_under("fused warp + cost volume L4", lambda: ops.warp_cost_volume(prv, nxt, flo), B * H * W * 81 * 2.0 * 32)
clock_under("WarpV2 L4", lambda: ops.warp(nxt, flo, "clamp"))
big = torch.empty(64 << 20, device=dev)
dst = torch.empty_like(big)
clock_under("256 MiB copy", lambda: dst.copy_(big), None, 10)
