#!/usr/bin/env python3
"""Segment stamps of sepconv3x3_pp_kernel (diagnostic build: make -C qpwcnet_amd/csrc ab ABSRC=optflow ABFLAGS=-DQPWC_SC_STAMP;
QPWC_HIP_LIB=.../libqpwc_ab.so): per stamped workgroup and half, shader cycles of every segment's work and barrier wait."""
import ctypes
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from qpwcnet_amd import _hip, ops  # noqa: E402

dev = "cuda:0"
g = torch.Generator(device=dev).manual_seed(0)
B, H, W = 8, 128, 256
layer = int(sys.argv[1]) if len(sys.argv) > 1 else 1
src_ch, F = {1: ((84, 32, 2), 128), 2: ((128,), 64)}[layer]
srcs = [torch.randn(B, H, W, c, device=dev, generator=g) for c in src_ch]
C = sum(src_ch)
dw = torch.randn(C, 9, device=dev, generator=g)
pw = ops.pad_pointwise(torch.randn(F, C, device=dev, generator=g) / C ** 0.5)
bias = torch.randn(F, device=dev, generator=g)
for _ in range(30):
    ops.sepconv3x3(srcs, dw, pw, bias, mish_on_store=True)
torch.cuda.synchronize()
buf = (ctypes.c_longlong * 256)()
L = _hip.lib()
L.qpwc_debug_sc_stamps.argtypes = [ctypes.POINTER(ctypes.c_longlong), ctypes.c_int]
assert L.qpwc_debug_sc_stamps(buf, 256) == 0
nsteps = (C + 31) // 32
nseg = 2 * nsteps + 1
for slot in range(4):
    st = [buf[slot * 64 + i] for i in range(1 + 4 * nseg)]
    if st[0] == 0:
        continue
    h = slot & 1
    out = []
    for sg in range(nseg):
        t_in, t_a, t_b, t_out = st[1 + 4 * sg], st[2 + 4 * sg], st[3 + 4 * sg], st[4 + 4 * sg]
        prev = st[4 * sg]
        tau = sg - h
        what = "idle" if tau < 0 or tau >= 2 * nsteps else ("OTH%d" % (tau >> 1) if tau % 2 == 0 else "MAT%d" % (tau >> 1))
        if what.startswith("OTH"):
            out.append("%s wait %d commit %d fetch %d depthwise %d" % (what, t_in - prev, t_a - t_in, t_b - t_a, t_out - t_b))
        else:
            out.append("%s wait %d work %d" % (what, t_in - prev, t_out - t_in))
    print("wg %d half %s total %d cycles | " % (slot >> 1, "AB"[h], st[4 * nseg] - st[0]) + " | ".join(out))
