#!/usr/bin/env python3
"""A/B in one process of how a single-process step keeps its EPE history (dist.EpeGather, no process group): the
side-stream copy of round 3 vs a clone on the compute stream (a ~4.6 us launch at the head of the next forward) vs no
copy at all (round 2: the history aliased two buffers).  bench.py's own loop (dist.timed_steps), B=8 256x512 fp32."""
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from qpwcnet_amd import dist as qdist, metrics, synth  # noqa: E402
from qpwcnet_amd.pwcnet import GraphedForward, build_flower  # noqa: E402

dev = torch.device("cuda", 0)
hw, B = (256, 512), 8
weights = synth.make_weights(42, hw)
pairs_np, gt_np = synth.make_frames(B, hw[0], hw[1], seed=1234)
pairs = torch.from_numpy(pairs_np).to(dev)
gt_pyr = metrics.multiscale_ground_truth(torch.from_numpy(gt_np).to(dev), [(hw[0] >> s, hw[1] >> s) for s in (5, 4, 3, 2, 1, 0)])
model = build_flower(True, hw, "channels_last", weights=weights, device=dev)


class CloneOnCompute(qdist.EpeGather):
    def submit(self, local_epe=None, slot=None):
        k = self.slot
        self.slot ^= 1
        self.pending.append((None, self.payload[k][:self.L]))

    def collect(self):
        _, buf = self.pending.pop(0)
        buf = buf.clone()
        return buf.unsqueeze(0), buf


class NoCopy(CloneOnCompute):
    def collect(self):
        _, buf = self.pending.pop(0)
        return buf.unsqueeze(0), buf


res = {}
for name, cls in (("side-stream copy", qdist.EpeGather), ("clone on the compute stream", CloneOnCompute), ("no copy", NoCopy)):
    gather = cls(6, dev, n_local=B)
    g0 = GraphedForward(model, pairs, epilogue=lambda fl: metrics.per_level_epe(gt_pyr, fl, out=gather.payload_view(0)), warmup=1)
    g1 = GraphedForward(model, pairs, epilogue=lambda fl: metrics.per_level_epe(gt_pyr, fl, out=gather.payload_view(1)), warmup=0,
                        share_with=g0)
    graphs = [g0, g1]

    def run_step(k):
        s = gather.next_slot()
        graphs[s].replay()
        return s
    res[name] = (gather, run_step)
for rnd in range(4):
    for name, (gather, run_step) in res.items():
        elapsed, results = qdist.timed_steps(run_step, gather, 50, 10, dev)
        print("%-28s round %d: %.4f ms/step" % (name, rnd, 1e3 * elapsed / 50), flush=True)
