#!/usr/bin/env python3
"""Steady-state per-kernel stats from a rocprofv3 --kernel-trace CSV of bench.py.
MIOpen's find-mode trials during warm-up dominate the raw --stats file; this keeps only
the forwards of the timed region (delimited by the epe_multi_final_kernel launch that ends
every forward).  usage: trace_steady.py <kernel_trace.csv> <first_fwd> <last_fwd> <out.csv>"""
import collections
import csv
import sys

path, lo, hi, out = sys.argv[1], int(sys.argv[2]), int(sys.argv[3]), sys.argv[4]
rows = list(csv.DictReader(open(path)))
rows.sort(key=lambda r: int(r["Start_Timestamp"]))
ends = [i for i, r in enumerate(rows) if "epe_multi_final_kernel" in r["Kernel_Name"]]
a, b = ends[lo - 1] + 1, ends[hi]
acc = collections.defaultdict(lambda: [0, 0])
for r in rows[a:b + 1]:
    k = r["Kernel_Name"]
    acc[k][0] += 1
    acc[k][1] += int(r["End_Timestamp"]) - int(r["Start_Timestamp"])
n = hi - lo + 1
span = int(rows[b]["End_Timestamp"]) - int(rows[a]["Start_Timestamp"])
tot = sum(v[1] for v in acc.values())
with open(out, "w") as f:
    w = csv.writer(f)
    w.writerow(["Name", "CallsPerForward", "AverageNs", "NsPerForward", "Percentage"])
    for k, (c, t) in sorted(acc.items(), key=lambda kv: -kv[1][1]):
        w.writerow([k, c / n, t / c, t / n, 100.0 * t / tot])
    w.writerow(["# forwards %d..%d: wall %.1f us/forward, kernel-busy %.1f us/forward, %d kernels/forward" % (
        lo, hi, span / n / 1e3, tot / n / 1e3, (b - a + 1) // n), "", "", "", ""])
print("wall us/forward", span / n / 1e3, "busy", tot / n / 1e3, "kernels/forward", (b - a + 1) / n)
