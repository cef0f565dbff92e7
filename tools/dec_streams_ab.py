#!/usr/bin/env python3
"""Timing experiment: which side stream each decoder level runs on (QpwcNet.dec_stream_of) -- whole forward + EPE
under hipGraph, B=8 256x512 fp32.  One mapping per PROCESS (a capture that the runtime refuses must not take the
others with it); the parent only starts children and prints their lines.

    python tools/dec_streams_ab.py                  # all mappings below, one child each
    python tools/dec_streams_ab.py 0,1,2,3          # one mapping in this process
"""
import os
import subprocess
import sys
import time

# ("0,1,0,1" -- two side streams waiting on each other alternately -- ended the child with SIGSEGV inside hipGraph
# capture when it was tried, once; QpwcNet now refuses such a mapping)
MAPPINGS = ["0,0,0,0", "0,1,1,1", "0,1,2,2", "0,1,2,3"]


def child(mapping):
    import torch
    sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
    from qpwcnet_amd import metrics, synth
    from qpwcnet_amd.pwcnet import GraphedForward, build_flower
    dev = "cuda:0"
    hw, B = (256, 512), 8
    weights = synth.make_weights(42, hw)
    pairs_np, gt_np = synth.make_frames(B, hw[0], hw[1], seed=1234)
    pairs = torch.from_numpy(pairs_np).to(dev)
    gt_pyr = metrics.multiscale_ground_truth(torch.from_numpy(gt_np).to(dev),
                                             [(hw[0] >> s, hw[1] >> s) for s in (5, 4, 3, 2, 1, 0)])
    model = build_flower(True, hw, "channels_last", weights=weights, device=dev)
    ref = [f.clone() for f in model(pairs)]
    model.dec_stream_of = tuple(int(x) for x in mapping.split(","))
    g = GraphedForward(model, pairs, epilogue=lambda fl: metrics.per_level_epe(gt_pyr, fl), warmup=2)
    outs, _ = g.replay()
    torch.cuda.synchronize()
    same = all(torch.equal(a, b) for a, b in zip(outs, ref))
    ts = []
    for _ in range(5):
        for _ in range(5):
            g.replay()
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        for _ in range(40):
            g.replay()
        torch.cuda.synchronize()
        ts.append((time.perf_counter() - t0) / 40 * 1e3)
    print("decoder levels on side streams %-8s flows identical to the eager forward: %s   ms/step %s   median %.4f" % (
        mapping, same, " ".join("%.4f" % t for t in ts), sorted(ts)[len(ts) // 2]), flush=True)


if __name__ == "__main__":
    if len(sys.argv) > 1:
        child(sys.argv[1])
    else:
        for m in MAPPINGS:
            p = subprocess.run([sys.executable, os.path.abspath(__file__), m], capture_output=True, text=True, timeout=240)
            line = [l for l in p.stdout.splitlines() if l.startswith("decoder levels")]
            print(line[0] if line else "mapping %s: rc %d %s" % (m, p.returncode, p.stderr.strip().splitlines()[-1:]), flush=True)
