#!/usr/bin/env python3
"""Which workgroups of sepconv3x3_fused_kernel share a CU, and in which phase each is when the other starts /
stores (diagnostic build -DQPWC_SC_STAMP, QPWC_HIP_LIB=.../libqpwc_ab.so): per workgroup start, first staged step,
last-step begin, end (s_memtime) + HW_ID / XCC_ID.  Prints, for the L4 first layer (115 -> 128), the placement
pattern and the phase offsets between co-resident workgroups."""
import collections
import ctypes
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from qpwcnet_amd import _hip, ops  # noqa: E402

dev = "cuda:0"
g = torch.Generator(device=dev).manual_seed(0)
B, H, W = 8, 128, 256
src_ch, F = (84, 32, 2), 128
srcs = [torch.randn(B, H, W, c, device=dev, generator=g) for c in src_ch]
C = sum(src_ch)
dw = torch.randn(C, 9, device=dev, generator=g)
pw = ops.pad_pointwise(torch.randn(F, C, device=dev, generator=g) / C ** 0.5)
bias = torch.randn(F, device=dev, generator=g)
for _ in range(10):
    ops.sepconv3x3(srcs, dw, pw, bias, mish_on_store=True)
torch.cuda.synchronize()
n = 2048
buf = (ctypes.c_longlong * (n * 6))()
L = _hip.lib()
L.qpwc_debug_sc_census.argtypes = [ctypes.POINTER(ctypes.c_longlong), ctypes.c_int]
assert L.qpwc_debug_sc_census(buf, n * 6) == 0
rec = [[buf[i * 6 + j] for j in range(6)] for i in range(n)]
t0 = min(r[0] for r in rec)
cus = collections.defaultdict(list)
for bid, r in enumerate(rec):
    hw, xcc = r[4], r[5] & 0xf
    cu, sh, se = (hw >> 8) & 0xf, (hw >> 12) & 1, (hw >> 13) & 0x7
    cus[(xcc, se, sh, cu)].append((r[0] - t0, r[1] - t0, r[2] - t0, r[3] - t0, bid))
print("distinct CUs:", len(cus), " workgroups per CU: min %d max %d" % (min(map(len, cus.values())), max(map(len, cus.values()))))
dur = sorted(r[3] - r[0] for r in rec)
print("workgroup lifetime cycles: median %d  p10 %d  p90 %d; kernel span %d cycles" % (dur[n // 2], dur[n // 10], dur[9 * n // 10], max(r[3] for r in rec) - t0))
# co-residency: for every workgroup, how far into its life is the workgroup that shares its CU when it starts
offs = []
for key, lst in cus.items():
    lst.sort()
    for i, a in enumerate(lst):
        for b in lst[:i]:
            if b[3] > a[0]:                      # b still running when a starts
                offs.append((a[0] - b[0]) / max(1, b[3] - b[0]))
offs.sort()
print("phase of the co-resident workgroup at a workgroup's start (fraction of its life): n=%d  p10 %.2f  median %.2f  p90 %.2f"
      % (len(offs), offs[len(offs) // 10], offs[len(offs) // 2], offs[9 * len(offs) // 10]))
hist = [0] * 10
for o in offs:
    hist[min(9, int(o * 10))] += 1
print("histogram by tenth of life:", hist)
k = sorted(cus)[5]
print("example CU", k, [(a[4], a[0], a[3]) for a in sorted(cus[k])][:8])
