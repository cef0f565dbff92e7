#!/usr/bin/env python3
"""A/B harness for the cost-volume kernels: checks three shapes (L4, L3, a ragged one) against a torch
restatement, then times 50 back-to-back launches five times.  The library reads its switches once per
process, so run it once per variant:

    python tools/cv_ab.py                                  # production build
    QPWC_HIP_LIB=.../libqpwc_exp.so QPWC_CV_PIPE=1 python tools/cv_ab.py     # make -C qpwcnet_amd/csrc experimental
    QPWC_STAMPS=1 ...                                      # also print the phase stamps of the experimental build
"""
import os, sys, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from qpwcnet_amd import ops

def ref(prv, nxt):
    B, H, W, C = prv.shape
    pad = torch.nn.functional.pad(nxt, (0, 0, 4, 4, 4, 4))
    outs = []
    for i in range(9):
        for j in range(9):
            outs.append((prv * pad[:, i:i + H, j:j + W]).mean(-1, keepdim=True))
    return torch.nn.functional.leaky_relu(torch.cat(outs, -1), 0.1)

g = torch.Generator(device="cuda").manual_seed(0)
for shape in [(8, 128, 256, 32), (8, 64, 128, 64), (3, 52, 76, 32)]:
    prv = torch.randn(*shape, device="cuda", generator=g); nxt = torch.randn(*shape, device="cuda", generator=g)
    got = ops.cost_volume(prv, nxt)
    err = float((got - ref(prv, nxt)).abs().max())
    for _ in range(5): ops.cost_volume(prv, nxt)
    ts = []
    for _ in range(5):
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        for _ in range(50): ops.cost_volume(prv, nxt)
        e1.record(); e1.synchronize(); ts.append(e0.elapsed_time(e1) / 50 * 1e3)
    print(shape, "max err %.2e" % err, "us med %.2f min %.2f" % (sorted(ts)[2], min(ts)), flush=True)

if os.environ.get("QPWC_STAMPS"):
    import ctypes
    from qpwcnet_amd import _hip
    L = _hip.lib()
    prv = torch.randn(8, 128, 256, 32, device="cuda"); nxt = torch.randn(8, 128, 256, 32, device="cuda")
    ops.cost_volume(prv, nxt); torch.cuda.synchronize()
    buf = (ctypes.c_uint64 * 96)()
    L.qpwc_debug_stamps.argtypes = [ctypes.c_void_p]
    print("rc", L.qpwc_debug_stamps(buf))
    names = [["A wait", "reads+mfma", "B wait", "frame write"], ["vmcnt+A wait", "dma issue", "epilogue", "B wait"]]
    for w in range(12):
        print("wave", w, "matrix" if w < 4 else ("loader" if w < 8 else "storer"),
              "  ".join("%s %d" % (names[w >= 4][k], buf[w * 8 + k]) for k in range(4)))
