#!/usr/bin/env python3
"""Per (kernel, grid size) launch statistics of the qpwc:: kernels in a rocprofv3 --kernel-trace CSV
(a kernel symbol runs at several pyramid levels; the per-symbol --stats average mixes them, weighted by
however many launches of each level the process happened to make).  The BackToBack columns cover only
launches that start within 3 us of the end of the previous launch of the same (kernel, grid) -- i.e.
bench.py's hipGraph of 50 replays of one launch, the figure its `roofline.avg_launch_ms` reports.
usage: by_grid.py <kernel_trace.csv> <out.csv> [skip_first_n_launches_per_key]"""
import collections
import csv
import statistics
import sys

path, out = sys.argv[1], sys.argv[2]
skip = int(sys.argv[3]) if len(sys.argv) > 3 else 0
rows = [r for r in csv.DictReader(open(path)) if "qpwc::" in r["Kernel_Name"]]
rows.sort(key=lambda r: int(r["Start_Timestamp"]))
acc = collections.defaultdict(list)
b2b = collections.defaultdict(list)
last_end = {}
for r in rows:
    key = (r["Kernel_Name"], int(r["Grid_Size_X"]))
    s, e = int(r["Start_Timestamp"]), int(r["End_Timestamp"])
    acc[key].append(e - s)
    if key in last_end and 0 <= s - last_end[key] < 3000:
        b2b[key].append(e - s)
    last_end[key] = e
with open(out, "w") as f:
    w = csv.writer(f)
    w.writerow(["Kernel", "Grid_Size_X", "Calls", "AverageNs", "MedianNs", "MinNs", "BackToBackCalls",
                "BackToBackAverageNs", "BackToBackMedianNs"])
    for (k, g), v in sorted(acc.items()):
        v = v[skip:] or v
        bb = b2b.get((k, g), [])
        w.writerow([k, g, len(v), "%.1f" % (sum(v) / len(v)), int(statistics.median(v)), min(v), len(bb),
                    "%.1f" % (sum(bb) / len(bb)) if bb else "", int(statistics.median(bb)) if bb else ""])
print("wrote", out, len(acc), "keys")
