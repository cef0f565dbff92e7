#!/usr/bin/env python3
"""SeparableConv2D of OptFlow at the level shapes: fused kernel
(qpwc_sepconv3x3_fwd) vs depthwise kernel + library GEMM, hipGraph replay of `iters` launches.

    python tools/sepbench.py [--batch 8] [--levels 2,3,4] [--iters 20]
"""
import argparse
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from qpwcnet_amd import ops  # noqa: E402


def timeit(fn, iters, rounds=5):
    for _ in range(3):
        fn()
    torch.cuda.synchronize()
    side = torch.cuda.Stream()
    g = torch.cuda.CUDAGraph()
    with torch.cuda.stream(side):
        with torch.cuda.graph(g, stream=side, capture_error_mode="thread_local"):
            for _ in range(iters):
                fn()
    torch.cuda.synchronize()
    ts = []
    for _ in range(rounds):
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        g.replay()
        e1.record()
        e1.synchronize()
        ts.append(e0.elapsed_time(e1) / iters * 1e3)
    return sorted(ts)[len(ts) // 2]


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--batch", type=int, default=8)
    ap.add_argument("--levels", default="2,3,4")
    ap.add_argument("--iters", type=int, default=20)
    ap.add_argument("--res", default="256x512", help="input resolution the level shapes derive from (1024x2048: config 4)")
    ap.add_argument("--fused-only", action="store_true")
    a = ap.parse_args()
    dev = "cuda:0"
    g = torch.Generator(device=dev).manual_seed(0)
    chans = [256, 256, 128, 64, 32]
    for l in map(int, a.levels.split(",")):
        rh, rw = (int(v) for v in a.res.split("x"))
        H, W, Cf = rh >> (5 - l), rw >> (5 - l), chans[l]
        B = a.batch
        layers = [((81, Cf, 2) if l > 0 else (81, Cf, Cf), 128), ((128,), 64), ((64,), 32), ((32,), 16)]
        for li, (src_ch, F) in enumerate(layers):
            C = sum(src_ch)
            srcs = [torch.randn(B, H, W, c, device=dev, generator=g) for c in src_ch]
            dw = torch.randn(C, 9, device=dev, generator=g)
            pw = torch.randn(F, C, device=dev, generator=g) / C ** 0.5
            bias = torch.randn(F, device=dev, generator=g)
            pwp = ops.pad_pointwise(pw)
            pwt = pw.t().contiguous()
            act = li > 0
            # inside a fused chain the producer stores Mish(z): layers 2..4 load activated tensors
            # and layers 1..3 activate at the store

            fsrcs, fdw, fpwp = srcs, dw, pwp
            if li == 0:  # the network feeds layer 1 an 84-channel cost volume (81 + 3 zero pads)
                z3 = torch.zeros(B, H, W, 3, device=dev)
                fsrcs = [torch.cat([srcs[0], z3], dim=3)] + srcs[1:]
                fdw = torch.cat([dw[:81], torch.zeros(3, 9, device=dev), dw[81:]]).contiguous()
                fpwp = ops.pad_pointwise(torch.cat([pw[:, :81], torch.zeros(F, 3, device=dev), pw[:, 81:]], dim=1))

            def fused():
                return ops.sepconv3x3(fsrcs, fdw, fpwp, bias, mish_on_load=False, mish_on_store=li < 3)

            def split():
                y = ops.dwconv3x3(srcs, dw, mish_on_load=act)
                return torch.addmm(bias, y.view(B * H * W, -1), pwt)

            if a.fused_only:
                err, tf, ts = float("nan"), timeit(fused, a.iters), float("nan")
            else:
                chk = ops.sepconv3x3(fsrcs, fdw, fpwp, bias, mish_on_load=act)
                err = float((chk.view(B * H * W, -1) - split()).abs().max())
                tf, ts = timeit(fused, a.iters), timeit(split, a.iters)
            flops = 2.0 * B * H * W * C * (F + 9)
            print("L%d layer %d  %dx%dx%d  C %3d -> F %3d : fused %7.1f us (%5.1f TF)   dw+gemm %7.1f us   max|diff| %.1e"
                  % (l, li + 1, B, H, W, C, F, tf, flops / tf * 1e-6, ts, err), flush=True)


if __name__ == "__main__":
    main()
