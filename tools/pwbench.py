#!/usr/bin/env python3
"""Pointwise half of the split first OptFlow layer at the two coarsest levels of config 2 (and L2 for scale): the own
matrix-core kernel (qpwc_pointwise_bias_fwd) vs the library GEMM (torch.addmm), hipGraph replay."""
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))
from qpwcnet_amd import ops  # noqa: E402
from sepbench import timeit  # noqa: E402

dev = "cuda:0"
g = torch.Generator(device=dev).manual_seed(0)
for name, M, C, F in (("L0", 8 * 8 * 16, 596, 128), ("L1", 8 * 16 * 32, 342, 128), ("L2", 8 * 32 * 64, 214, 128)):
    y = torch.randn(M, C, device=dev, generator=g)
    w = torch.randn(F, C, device=dev, generator=g) / C ** 0.5
    b = torch.randn(F, device=dev, generator=g)
    wp, wt = ops.pad_pointwise(w), w.t().contiguous()
    t_own = timeit(lambda: ops.pointwise_bias(y, wp, b), 20)
    t_lib = timeit(lambda: torch.addmm(b, y, wt), 20)
    print("%s  M %5d  C %3d -> F %3d : own %6.1f us   library GEMM %6.1f us   max|diff| %.1e"
          % (name, M, C, F, t_own, t_lib, float((ops.pointwise_bias(y, wp, b) - torch.addmm(b, y, wt)).abs().max())), flush=True)
