#!/usr/bin/env python3
"""Writes tests/golden/*.npz from the float64 numpy oracle (oracle/np_ref.py).

Run from the repo root:  python tests/golden/make_golden.py
Each file holds: sampled flat output indices + float64 values, the float64 sum
and abs-sum of the whole output, and the case description.  Inputs are NOT
stored; they are regenerated from the seed by tests/golden/cases.py.
"""
import os
import sys

import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, os.path.join(HERE, "..", ".."))
sys.path.insert(0, HERE)

from oracle import np_ref  # noqa: E402
import cases  # noqa: E402


def run_oracle(name, dtype=np.float64):
    op, fmt, shape, seed, extra = cases.CASES[name]
    a, b = cases.make_inputs(name, dtype)
    if op == "cost_volume":
        return np_ref.cost_volume(a, b, extra.get("search_range", 4), fmt)
    if op == "warp_v2":
        return np_ref.warp_v2(a, b, fmt)
    return np_ref.tf_warp(a, b, fmt)


def main():
    for name in cases.CASES:
        out = run_oracle(name)
        idx = cases.sample_indices(name, out.size)
        flat = out.reshape(-1)
        np.savez_compressed(
            os.path.join(HERE, name + ".npz"),
            idx=idx.astype(np.int64), val=flat[idx].astype(np.float64),
            total=np.float64(flat.sum()), abs_total=np.float64(np.abs(flat).sum()),
            shape=np.asarray(out.shape, np.int64))
        print(name, out.shape, float(np.abs(flat).sum()))

    # known-answer case of qpwcnet/app/optical_flow/test_warp.py:28-33 (full tensors)
    nxt = np.float32([[0, 0, 0], [0, 1, 0], [0, 0, 0]]).reshape(1, 3, 3, 1)
    flo = np.float32([1, 0]).reshape(1, 1, 1, 2)
    np.savez_compressed(os.path.join(HERE, "known_3x3.npz"), nxt=nxt, flo=flo,
                        warp_v2=np_ref.warp_v2(nxt, flo), tf_warp=np_ref.tf_warp(nxt, flo))

    # inverse flow / occlusion maps (fp32 by definition: occlusion.py:56-57 casts to float32):
    # full uint8 map + the inverse flow
    for name in cases.OCC_CASES:
        fmt = cases.OCC_CASES[name][0]
        flow = cases.make_flow(name)
        occ = np_ref.estimate_occlusion_map(flow, fmt)
        inv = np_ref.invert_flow(flow, fmt)
        assert set(np.unique(occ)) <= {0.0, 1.0}
        np.savez_compressed(os.path.join(HERE, name + ".npz"), occ=occ.astype(np.uint8),
                            inv=inv.astype(np.float32))
        print(name, occ.shape, float(occ.mean()))


if __name__ == "__main__":
    main()
