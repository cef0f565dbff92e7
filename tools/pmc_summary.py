#!/usr/bin/env python3
"""Summarise tools/pmc.sh output: per kernel name, mean counter value per dispatch."""
import collections
import csv
import glob
import os
import sys

root = sys.argv[1]
filt = sys.argv[2] if len(sys.argv) > 2 else ""
acc = collections.defaultdict(lambda: collections.defaultdict(lambda: [0, 0.0]))
for f in glob.glob(os.path.join(root, "*", "**", "*counter_collection.csv"), recursive=True):
    for r in csv.DictReader(open(f)):
        k = r["Kernel_Name"]
        if filt and not any(a in k for a in filt.split("|")):
            continue
        a = acc[k[:70]][r["Counter_Name"]]
        a[0] += 1
        a[1] += float(r["Counter_Value"])
for k, d in acc.items():
    print(k)
    for c, (n, v) in sorted(d.items()):
        print("   %-34s n=%4d mean=%16.1f" % (c, n, v / n))
