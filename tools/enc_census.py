#!/usr/bin/env python3
"""Where a workgroup of the narrow encoder kernel spends its life, and how the 4096 workgroups of a launch line up in
time (diagnostic build: make -C qpwcnet_amd/csrc ab ABSRC=encoder ABFLAGS=-DQPWC_ENC_STAMP; QPWC_HIP_LIB=.../libqpwc_ab.so).
Per workgroup: start, inputs landed, staged, matrix work done, end (s_memtime) + HW_ID."""
import collections
import ctypes
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from qpwcnet_amd import _hip, ops  # noqa: E402

dev = "cuda:0"
g = torch.Generator(device=dev).manual_seed(0)
C, H, W = 16, 128, 256
x = torch.randn(16, H, W, C, device=dev, generator=g)
w = (torch.randn(C, C, 3, 3, device=dev, generator=g) / (9 * C) ** 0.5).contiguous(memory_format=torch.channels_last)
b = torch.randn(C, device=dev, generator=g)
taps = ops.conv3x3_taps(w)
for _ in range(10):
    ops.conv3x3_mish(x, taps, b)
torch.cuda.synchronize()
n = 4096
buf = (ctypes.c_longlong * (n * 6))()
L = _hip.lib()
L.qpwc_debug_enc_census.argtypes = [ctypes.POINTER(ctypes.c_longlong), ctypes.c_int]
assert L.qpwc_debug_enc_census(buf, n * 6) == 0
rec = [[buf[i * 6 + j] for j in range(6)] for i in range(n)]
# s_memtime is per XCD (unsynchronised bases): workgroup i runs on XCD i % 8 -- normalise per XCD
by_xcd = collections.defaultdict(list)
for i in range(n):
    hw = rec[i][5] & 0xffffffff   # clocks are only comparable inside one CU: key = (XCC, SE, SH, CU)
    by_xcd[((rec[i][5] >> 32) & 0xf, (hw >> 13) & 0x7, (hw >> 12) & 1, (hw >> 8) & 0xf)].append(i)
print("CUs seen: %d, workgroups per CU: min %d max %d" % (len(by_xcd), min(map(len, by_xcd.values())), max(map(len, by_xcd.values()))))
for xcd, idx in by_xcd.items():
    base = min(rec[i][0] for i in idx)
    for i in idx:
        for j in range(5):
            rec[i][j] -= base
t0 = 0
span = max(r[4] for r in rec)
print("kernel span %d cycles (s_memtime ticks, per-XCD start = 0)" % span)
names = ["inputs landed", "staged (LDS write + barrier)", "matrix work", "Mish + stores issued"]
for j, nm in enumerate(names):
    d = sorted(r[j + 1] - r[j] for r in rec)
    print("  %-30s median %6d  p10 %6d  p90 %6d" % (nm, d[n // 2], d[n // 10], d[9 * n // 10]))
life = sorted(r[4] - r[0] for r in rec)
print("  %-30s median %6d  p10 %6d  p90 %6d" % ("workgroup lifetime", life[n // 2], life[n // 10], life[9 * n // 10]))
# how many workgroups are in which phase over time (20 slices of the span)
print("time slice: workgroups loading / staging / in matrix work / storing / resident")
for sl in range(20):
    t = t0 + span * (sl + 0.5) / 20
    cnt = [0, 0, 0, 0]
    for r in rec:
        for j in range(4):
            if r[j] <= t < r[j + 1]:
                cnt[j] += 1
    print("  %2d: %5d %5d %5d %5d   %5d" % (sl, cnt[0], cnt[1], cnt[2], cnt[3], sum(cnt)))
cus = collections.Counter()
for r in rec:
    hw = r[5]
    cus[((hw >> 8) & 0xf, (hw >> 12) & 1, (hw >> 13) & 0x7)] += 1
starts = sorted(r[0] - t0 for r in rec)
print("start times: p25 %d  median %d  p75 %d  last %d" % (starts[n // 4], starts[n // 2], starts[3 * n // 4], starts[-1]))
