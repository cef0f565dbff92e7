#!/usr/bin/env python3
"""Per-level EPE against the CPU oracle (oracle/net_ref.py) of the fp32-instruction path and of the bf16x3 path, and the
two paths against each other: 2 pairs at 256x512 (test infrastructure: the oracle is the checker here, as in tests/)."""
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from oracle import net_ref, torch_ref  # noqa: E402
from qpwcnet_amd import synth  # noqa: E402
from qpwcnet_amd.pwcnet import build_flower  # noqa: E402

hw = (256, 512)
weights = synth.make_weights(42, hw)
pairs, _ = synth.make_frames(2, hw[0], hw[1], seed=1234)
model = build_flower(True, hw, "channels_last", weights=weights, device="cuda:0")
f32 = [f.cpu() for f in model.predict(pairs)]
model.matmul = "bf16x3"
x3 = [f.cpu() for f in model.predict(pairs)]
ref = net_ref.RefNet(weights)(pairs)
for lvl, (a, b, r) in enumerate(zip(f32, x3, ref)):
    d = (a - b)
    print("level %d: EPE vs oracle  fp32 instructions %.2e   bf16x3 %.2e   | between the two: mean |d| %.2e  max |d| %.2e  (|flow| max %.2f)"
          % (lvl, float(torch_ref.epe_error(a, r)), float(torch_ref.epe_error(b, r)), float(d.abs().mean()), float(d.abs().max()),
             float(a.abs().max())), flush=True)
