#!/usr/bin/env python3
"""Phase stamps of the role-split SeparableConv2D kernel (diagnostic build: make -C qpwcnet_amd/csrc ab
ABSRC=optflow ABFLAGS="-DQPWC_SC_WS=64 -DQPWC_SC_STAMP"; run with QPWC_HIP_LIB=.../libqpwc_ab.so).  Launches the
first (layer 1) or second L4 OptFlow layer back to back and prints, for four stamped workgroups, the shader
cycles between the stamps of matrix wave 0 and of vector wave 4."""
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
cfg = {1: ((84, 32, 2), 128), 2: ((128,), 64)}[layer]
src_ch, F = cfg
srcs = [torch.randn(B, H, W, c, device=dev, generator=g) for c in src_ch]
C = sum(src_ch)
dw = torch.randn(C, 9, device=dev, generator=g)
pw = ops.pad_pointwise(torch.randn(F, C, device=dev, generator=g) / C ** 0.5)
bias = torch.randn(F, device=dev, generator=g)
for _ in range(30):
    ops.sepconv3x3(srcs, dw, pw, bias, mish_on_load=False, mish_on_store=True)
torch.cuda.synchronize()
buf = (ctypes.c_longlong * 256)()
L = _hip.lib()
L.qpwc_debug_sc_stamps.argtypes = [ctypes.POINTER(ctypes.c_longlong), ctypes.c_int]
assert L.qpwc_debug_sc_stamps(buf, 256) == 0
nsteps = (C + 31) // 32
mn = ["start", "S0", "S1"]
vn = ["start", "S0", "dw0", "S1"]
for k in range(nsteps - 1):
    mn += ["h0_%d" % k, "Sa%d" % k, "h1_%d" % k, "Sb%d" % k]
    vn += ["Sa%d" % k, "dw%d" % (k + 1), "Sb%d" % k]
rounds = max(F // 32, 1)
mn += ["Sa_last"] + ["r%d" % r for r in range(rounds)] + ["end"]
vn += ["Sa_last", "epilogue"]
for wgi in range(4):
    for role, names, off in (("matrix", mn, 0), ("vector", vn, 32)):
        st = [buf[wgi * 64 + off + i] for i in range(len(names))]
        if st[0] == 0:
            continue
        t0 = buf[wgi * 64]
        print("wg %d %s (+%d) total %d: " % (wgi, role, st[0] - t0, st[-1] - st[0])
              + "  ".join("%s %d" % (n, st[i] - st[i - 1]) for i, n in enumerate(names) if i > 0))
