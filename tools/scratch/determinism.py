"""Run-to-run bit stability of the fp32 matrix-core kernels under contention (two / three waves per SIMD issuing matrix
instructions): any difference between repeated launches on the same inputs is a hardware hazard in the generated code."""
import os, sys
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
from qpwcnet_amd import ops
DEV = "cuda:0"
N = int(sys.argv[1]) if len(sys.argv) > 1 else 200
g = torch.Generator(device=DEV).manual_seed(0)
def rnd(*s): return torch.randn(*s, device=DEV, generator=g)

def check(name, fn):
    ref = fn().clone()
    bad = 0
    worst = 0.0
    for i in range(N):
        out = fn()
        if not torch.equal(out, ref):
            bad += 1
            worst = max(worst, float((out - ref).abs().max()))
    print("%-60s %d launches, %d differ from the first%s" % (name, N, bad, "" if not bad else "  max|diff| %.3g" % worst), flush=True)
    return bad

total = 0
B, H, W = 8, 128, 256
prv, nxt, flo = rnd(B, H, W, 32), rnd(B, H, W, 32), rnd(B, H, W, 2) * 3
total += check("cost volume L4 (lds kernel)", lambda: ops.cost_volume(prv, nxt))
total += check("fused warp + cost volume L4 (8x16 regions)", lambda: ops.warp_cost_volume(prv, nxt, flo))
p3, n3, f3 = rnd(B, 64, 128, 64), rnd(B, 64, 128, 64), rnd(B, 64, 128, 2) * 3
total += check("cost volume L3", lambda: ops.cost_volume(p3, n3))
total += check("fused warp + cost volume L3", lambda: ops.warp_cost_volume(p3, n3, f3))
p1, n1 = rnd(B, 16, 32, 256), rnd(B, 16, 32, 256)
total += check("cost volume L1 (split-K kernel)", lambda: ops.cost_volume(p1, n1))
for (chans, F) in (((84, 32, 2), 128), ((128,), 64), ((64,), 32), ((32,), 16)):
    C = sum(chans)
    srcs = [rnd(B, H, W, c) for c in chans]
    dw, pw, bias = rnd(C, 9), rnd(F, C) / C ** 0.5, rnd(F)
    pwp = ops.pad_pointwise(pw)
    total += check("sepconv3x3 L4 %s -> %d" % (chans, F), lambda: ops.sepconv3x3(srcs, dw, pwp, bias, mish_on_store=True))
for C in (16, 32, 64, 128, 256):
    hw = {16: (128, 256), 32: (64, 128), 64: (32, 64), 128: (16, 32), 256: (8, 16)}[C]
    x = rnd(16, hw[0], hw[1], C)
    taps = ops.conv3x3_taps(rnd(C, C, 3, 3) / (3 * C ** 0.5))
    bias = rnd(C)
    total += check("encoder conv3x3 C=%d" % C, lambda: ops.conv3x3_mish(x, taps, bias))
print("TOTAL differing launches:", total)
