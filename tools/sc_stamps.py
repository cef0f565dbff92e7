#!/usr/bin/env python3
"""Phase stamps of sepconv3x3_fused_kernel (diagnostic build: make -C qpwcnet_amd/csrc ab ABSRC=optflow
ABFLAGS=-DQPWC_SC_STAMP; run with QPWC_HIP_LIB=.../libqpwc_ab.so).  Launches the first L4 OptFlow layer
(115 -> 128) back to back and prints, for four stamped workgroups, the shader cycles between stamps."""
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
cfg = {1: ((84, 32, 2), 128, False, True), 2: ((128,), 64, False, True), 3: ((64,), 32, False, True), 4: ((32,), 16, False, False)}[layer]
src_ch, F, act, act_out = cfg
srcs = [torch.randn(B, H, W, c, device=dev, generator=g) for c in src_ch]
C = sum(src_ch)
dw = torch.randn(C, 9, device=dev, generator=g)
pw = ops.pad_pointwise(torch.randn(F, C, device=dev, generator=g) / C ** 0.5)
bias = torch.randn(F, device=dev, generator=g)
for _ in range(30):
    ops.sepconv3x3(srcs, dw, pw, bias, mish_on_load=act, mish_on_store=act_out)
torch.cuda.synchronize()
buf = (ctypes.c_longlong * 256)()
L = _hip.lib()
L.qpwc_debug_sc_stamps.argtypes = [ctypes.POINTER(ctypes.c_longlong), ctypes.c_int]
rc = L.qpwc_debug_sc_stamps(buf, 256)
assert rc == 0, rc
names = ["start", "staged0", "dw0"]
nsteps = (C + 31) // 32
for k in range(nsteps - 1):
    names += ["A%d" % k, "B%d" % k, "pw%d+dw%d" % (k, k + 1)]
names += ["A_last", "B_last", "pw_last(+epi)", "last_epi"]
for wgi in range(4):
    st = [buf[wgi * 64 + i] for i in range(len(names))]
    if st[0] == 0:
        continue
    d = [st[i] - st[i - 1] for i in range(1, len(st))]
    print("wg %d total %d cycles: " % (wgi, st[-1] - st[0]) + "  ".join("%s %d" % (n, x) for n, x in zip(names[1:], d)))
