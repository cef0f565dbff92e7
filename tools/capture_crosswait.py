#!/usr/bin/env python3
"""Cause of the hipGraph-capture SIGSEGV of rounds 1 / 3 (decoder levels on two side streams that wait on each other
alternately, QpwcNet.dec_stream_of = (0,1,0,1)): each case ONCE, in a child process of its own with faulthandler on
(the Python frame of a host SIGSEGV goes to the child's stderr), the parent only collects.

  A  plain torch, no qpwc kernels: 4 "levels" alternating between two side streams, each waiting on the event of the
     level before (recorded on the OTHER side stream), outputs allocated under capture, Tensor.record_stream on them
  B  the same without record_stream
  C  the model, mapping (0,1,0,1), record_stream skipped while capturing
  D  the model, mapping (0,1,0,1), as in round 3 (record_stream under capture)

    python tools/capture_crosswait.py            # all four, one child each
    python tools/capture_crosswait.py C          # one case in this process
"""
import faulthandler
import os
import subprocess
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def plain(note):
    import torch
    dev = "cuda:0"
    x = torch.randn(1 << 20, device=dev)
    sides = [torch.cuda.Stream(), torch.cuda.Stream()]

    def forward():
        main = torch.cuda.current_stream()
        for sd in sides:
            sd.wait_stream(main)
        ready, outs, f = [], [], x
        acc = x * 0.5                                   # "flow level 0" on main
        for i in range(4):
            side = sides[i % 2]
            if i > 0:
                side.wait_event(ready[i - 1])           # recorded on the other side stream
            with torch.cuda.stream(side):
                f = f * 1.25 + float(i)                 # allocated on `side` (under capture: the private pool)
                if note:
                    for sd in [main] + sides:
                        if sd is not side:
                            f.record_stream(sd)
                ev = torch.cuda.Event()
                ev.record(side)
                ready.append(ev)
                outs.append(f)
            main.wait_event(ready[i])                   # "flow level i + 1" consumes decoder level i
            acc = acc + outs[i]
        return acc

    ref = forward()
    torch.cuda.synchronize()
    print("eager ok", flush=True)
    g = torch.cuda.CUDAGraph()
    cs = torch.cuda.Stream()
    cs.wait_stream(torch.cuda.current_stream())
    with torch.cuda.stream(cs):
        forward()
    torch.cuda.current_stream().wait_stream(cs)
    with torch.cuda.graph(g, capture_error_mode="thread_local"):
        out = forward()
    print("captured + instantiated", flush=True)
    g.replay()
    torch.cuda.synchronize()
    print("replayed; equal to eager:", bool(torch.equal(out, ref)), flush=True)


def model_case(note):
    import torch
    sys.path.insert(0, ROOT)
    from qpwcnet_amd import synth
    from qpwcnet_amd.pwcnet import GraphedForward, build_flower
    dev = "cuda:0"
    hw, B = (256, 512), 8
    weights = synth.make_weights(42, hw)
    pairs_np, _ = synth.make_frames(B, hw[0], hw[1], seed=1234)
    pairs = torch.from_numpy(pairs_np).to(dev)
    model = build_flower(True, hw, "channels_last", weights=weights, device=dev)
    with torch.no_grad():
        ref = [f.clone() for f in model(pairs)]
    model.allow_returning_dec_streams = True
    model.record_stream_under_capture = note
    model.dec_stream_of = (0, 1, 0, 1)
    with torch.no_grad():
        eager = model(pairs)
    torch.cuda.synchronize()
    print("eager (0,1,0,1) ok; equal to the default mapping:", all(torch.equal(a, b) for a, b in zip(eager, ref)), flush=True)
    g = GraphedForward(model, pairs, warmup=1)
    print("captured + instantiated", flush=True)
    outs, _ = g.replay()
    torch.cuda.synchronize()
    print("replayed; equal to the default mapping:", all(torch.equal(a, b) for a, b in zip(outs, ref)), flush=True)


CASES = {"A": lambda: plain(True), "B": lambda: plain(False), "C": lambda: model_case(False), "D": lambda: model_case(True)}

if __name__ == "__main__":
    if len(sys.argv) > 1:
        faulthandler.enable(all_threads=True)
        CASES[sys.argv[1]]()
    else:
        for c in "ABCD":
            p = subprocess.run([sys.executable, "-X", "faulthandler", os.path.abspath(__file__), c], capture_output=True,
                               text=True, timeout=300)
            print("== case %s: rc %d" % (c, p.returncode), flush=True)
            print(p.stdout.strip(), flush=True)
            if p.returncode != 0:
                print("-- stderr tail:\n" + "\n".join(p.stderr.strip().splitlines()[-40:]), flush=True)
