#!/usr/bin/env python3
"""How long is ONE L4 cost-volume launch (8x128x256x32 fp32, 84-float pixels)?  The same launch timed
four ways with HIP events; run it under `rocprofv3 --kernel-trace` and tools/by_grid.py to see the
profiler's per-dispatch durations of the very same launches (phases are separated by 1 ms sleeps).

  A  hipGraph of 50 launches, events around the replay          (bench.py's avg_launch_ms)
  B  50 eager launches back to back, events around all
  C  50 eager launches, each between its own pair of events, device synchronised before each launch
  D  as C but a different kernel (L4 WarpV2) runs before each launch (what the step does)
"""
import os
import sys
import time

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from qpwcnet_amd import ops  # noqa: E402

dev = "cuda"
g = torch.Generator(device=dev).manual_seed(0)
prv = torch.randn(8, 128, 256, 32, device=dev, generator=g)
nxt = torch.randn(8, 128, 256, 32, device=dev, generator=g)
flo = torch.randn(8, 128, 256, 2, device=dev, generator=g) * 4
buf = torch.empty(8, 128, 256, 84, device=dev)
N = 50


def fn():
    ops.cost_volume_into(prv, nxt, buf, 0)


def ev():
    return torch.cuda.Event(enable_timing=True)


for _ in range(5):
    fn()
torch.cuda.synchronize()

gr, side = torch.cuda.CUDAGraph(), torch.cuda.Stream()
with torch.cuda.stream(side):
    with torch.cuda.graph(gr, stream=side, capture_error_mode="thread_local"):
        for _ in range(N):
            fn()
torch.cuda.synchronize()
res = {}
ts = []
for _ in range(5):
    e0, e1 = ev(), ev()
    e0.record(); gr.replay(); e1.record(); e1.synchronize()
    ts.append(e0.elapsed_time(e1) / N * 1e3)
res["A graph of 50"] = sorted(ts)[2]
time.sleep(0.001)
ts = []
for _ in range(5):
    e0, e1 = ev(), ev()
    e0.record()
    for _ in range(N):
        fn()
    e1.record(); e1.synchronize()
    ts.append(e0.elapsed_time(e1) / N * 1e3)
res["B eager back to back"] = sorted(ts)[2]
time.sleep(0.001)
ts = []
for _ in range(N):
    torch.cuda.synchronize()
    e0, e1 = ev(), ev()
    e0.record(); fn(); e1.record(); e1.synchronize()
    ts.append(e0.elapsed_time(e1) * 1e3)
res["C isolated, own events"] = sorted(ts)[N // 2]
time.sleep(0.001)
ts = []
for _ in range(N):
    torch.cuda.synchronize()
    ops.warp(nxt, flo, "clamp")
    e0, e1 = ev(), ev()
    e0.record(); fn(); e1.record(); e1.synchronize()
    ts.append(e0.elapsed_time(e1) * 1e3)
res["D after a warp launch, own events"] = sorted(ts)[N // 2]
for k, v in res.items():
    print("%-36s %7.2f us" % (k, v))
