#!/usr/bin/env python3
"""Phase stamps of cost_volume_mfma_lds_kernel (diagnostic build: make -C qpwcnet_amd/csrc ab
ABFLAGS=-DQPWC_CV_STAMP; run with QPWC_HIP_LIB=.../libqpwc_ab.so): the L4 launch of the bench
(8x128x256x32, 84-float pixels) back to back, shader cycles between the stamps of wave 0 of 8 workgroups."""
import ctypes
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from qpwcnet_amd import _hip, ops  # noqa: E402

dev = "cuda:0"
g = torch.Generator(device=dev).manual_seed(0)
shape = (8, 128, 256, 32)
prv = torch.randn(*shape, device=dev, generator=g)
nxt = torch.randn(*shape, device=dev, generator=g)
flo = torch.randn(*shape[:3], 2, device=dev, generator=g) * 4
buf = torch.empty(shape[:3] + (84,), device=dev)
fused = len(sys.argv) > 1 and sys.argv[1] == "fused"
for _ in range(40):
    ops.cost_volume_into(prv, nxt, buf, 0, flo=flo if fused else None)
torch.cuda.synchronize()
out = (ctypes.c_longlong * 128)()
L = _hip.lib()
L.qpwc_debug_cv_stamps.argtypes = [ctypes.POINTER(ctypes.c_longlong), ctypes.c_int]
assert L.qpwc_debug_cv_stamps(out, 128) == 0
names = ["start", "loads_issued", "loads_landed+lds_write", "barrier", "mfma_issued", "barrier2", "frame_written", "readback+stores_issued", "stores_acked"]
for w in range(8):
    st = [out[w * 16 + i] for i in range(len(names))]
    if st[0] == 0:
        continue
    print("wg %d total %6d: " % (w, st[-1] - st[0]) + "  ".join("%s %d" % (n, st[i + 1] - st[i]) for i, n in enumerate(names[1:])))
