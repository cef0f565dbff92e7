#!/usr/bin/env python3
"""ms/step of one hipGraph of the forward + EPE (B=8, 256x512 fp32), model attributes set from the command line
(name=value, python literals) -- for A/Bs across PROCESSES (HIP runtime environment knobs, builds via QPWC_HIP_LIB):
    GPU_MAX_HW_QUEUES=8 python tools/step_time.py
    python tools/step_time.py "dec_stream_of=(0,1,1,1)"
    python tools/step_time.py --batch=32 --dtype=f16 "dec_chunks=(1,2,2,2)"      # BASELINE config 5's workload """
import ast
import os
import sys
import time

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from qpwcnet_amd import metrics, synth  # noqa: E402
from qpwcnet_amd.pwcnet import GraphedForward, build_flower  # noqa: E402

dev = "cuda:0"
hw, B, tdtype = (256, 512), 8, torch.float32
for a in [a for a in sys.argv[1:] if a.startswith("--")]:      # --batch=32 --dtype=f16 --hw=1024x2048
    sys.argv.remove(a)
    k, v = a[2:].split("=", 1)
    if k == "batch":
        B = int(v)
    elif k == "dtype":
        tdtype = torch.float16 if v == "f16" else torch.float32
    elif k == "hw":
        hw = tuple(int(x) for x in v.split("x"))
weights = synth.make_weights(42, hw)
pairs_np, gt_np = synth.make_frames(B, hw[0], hw[1], seed=1234)
pairs = torch.from_numpy(pairs_np).to(dev, tdtype)
gt_pyr = metrics.multiscale_ground_truth(torch.from_numpy(gt_np).to(dev), [(hw[0] >> s, hw[1] >> s) for s in (5, 4, 3, 2, 1, 0)])
model = build_flower(True, hw, "channels_last", weights=weights, device=dev, dtype=tdtype)
for a in sys.argv[1:]:
    k, v = a.split("=", 1)
    if "." in k or "[" in k:      # a dotted path below the model, e.g. "upflows[0].flow.fused_sepconv=True"
        exec("model.{} = ast.literal_eval(v)".format(k))
    else:
        setattr(model, k, ast.literal_eval(v))
g = GraphedForward(model, pairs, epilogue=lambda fl: metrics.per_level_epe(gt_pyr, fl), warmup=2)
ts = []
for _ in range(6):
    for _ in range(5):
        g.replay()
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(40):
        g.replay()
    torch.cuda.synchronize()
    ts.append((time.perf_counter() - t0) / 40 * 1e3)
tag = " ".join("%s=%s" % (k, os.environ[k]) for k in sorted(os.environ) if k.startswith(("DEBUG_", "GPU_", "ROC_", "AMD_", "QPWC_")))
print("%-60s %s   median %.4f ms/step" % (tag + " " + " ".join(sys.argv[1:]), " ".join("%.4f" % t for t in ts), sorted(ts)[len(ts) // 2]), flush=True)
