#!/usr/bin/env python3
"""Kernel timeline of ONE steady-state forward from tools/ktrace.py's timeline.csv: the shortest forward
with the modal kernel count; start/duration/queue/grid/name per kernel, plus the idle gaps of every queue.
usage: fwd_timeline.py timeline.csv [out.txt]"""
import collections
import csv
import sys

rows = list(csv.DictReader(open(sys.argv[1])))
ends = [i for i, r in enumerate(rows) if "epe_multi_final" in r["name"]]
counts = collections.Counter(b - a for a, b in zip(ends[:-1], ends[1:]))
modal = counts.most_common(1)[0][0]
best = None
for a, b in zip(ends[:-1], ends[1:]):
    if b - a == modal:
        w = float(rows[b]["start_us"]) + float(rows[b]["dur_us"]) - float(rows[a + 1]["start_us"])
        if best is None or w < best[0]:
            best = (w, a, b)
w, a, b = best
out = open(sys.argv[2], "w") if len(sys.argv) > 2 else sys.stdout
out.write("# %d kernels, wall %.1f us (shortest of %d forwards with the modal kernel count)\n" % (modal, w, counts[modal]))
t0 = float(rows[a + 1]["start_us"])
last = {}
for r in rows[a + 1:b + 1]:
    s, d, q = float(r["start_us"]) - t0, float(r["dur_us"]), r["queue"]
    gap = s - last.get(q, s)
    out.write("%8.1f %7.1f  q%s gap %6.1f  g%-8s %s\n" % (s, d, q, gap, r["grid"], r["name"][:72]))
    last[q] = s + d
