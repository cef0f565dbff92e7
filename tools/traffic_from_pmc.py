#!/usr/bin/env python3
"""profiles/traffic.json from a tools/pmc.sh run: HBM bytes per launch of the hot-path kernels, from the
FETCH_SIZE / WRITE_SIZE passes (separate --pmc runs).

Corrections per MI355X_MICROARCH.md (HBM section): both counters are in KiB; on gfx950 FETCH_SIZE reports
exactly half of the bytes of a wide (16 B/lane) coalesced streaming read -> doubled; WRITE_SIZE is exact for
streaming stores.  The file is stamped with the sha256 of the kernel sources (qpwcnet_amd._hip.source_sha256):
bench.py drops `roofline.traffic` when the sources have changed since.

usage: traffic_from_pmc.py <pmc dir> <out.json> <tag> <key>=<kernel substring>[|<alternative>...] [...]
An existing <out.json> with the same source hash and tag is extended (one call per profiled shape)."""
import csv
import glob
import json
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from qpwcnet_amd import _hip  # noqa: E402

root, out, tag = sys.argv[1:4]
pairs = [a.split("=", 1) for a in sys.argv[4:]]
rows = []
for f in glob.glob(os.path.join(root, "*", "**", "*counter_collection.csv"), recursive=True):
    rows += list(csv.DictReader(open(f)))
prev = {}
if os.path.exists(out):
    try:
        prev = json.load(open(out))
    except Exception:  # noqa: BLE001
        prev = {}
    if prev.get("kernel_source_sha256") != _hip.source_sha256() or prev.get("tag") != tag:
        prev = {}
d = {"kernel_source_sha256": _hip.source_sha256(), "tag": tag,
     "note": "separate rocprofv3 --pmc FETCH_SIZE / --pmc WRITE_SIZE passes over tools/cv84_launch.py "
             "(tools/make_traffic.sh); FETCH_SIZE KiB x2 (gfx950 16-B/lane correction), WRITE_SIZE KiB; the "
             "cost-volume launches write 84-float pixels (81 channels + 3 zeroed pads), the algorithmic bytes "
             "count 81"}
d.update({k: v for k, v in prev.items() if k not in d})
for key, sub in pairs:
    vals = {"FETCH_SIZE": [], "WRITE_SIZE": []}
    for r in rows:
        if any(a in r["Kernel_Name"] for a in sub.split("|")) and r["Counter_Name"] in vals:
            vals[r["Counter_Name"]].append(float(r["Counter_Value"]))
    if not vals["FETCH_SIZE"] or not vals["WRITE_SIZE"]:
        print("no counters for", key, sub)
        continue
    fetch = sum(vals["FETCH_SIZE"]) / len(vals["FETCH_SIZE"]) * 1024 * 2
    write = sum(vals["WRITE_SIZE"]) / len(vals["WRITE_SIZE"]) * 1024
    d[key] = fetch + write
    d[key + "_detail"] = {"kernel": sub, "launches": len(vals["FETCH_SIZE"]),
                          "FETCH_SIZE_KiB_raw": sum(vals["FETCH_SIZE"]) / len(vals["FETCH_SIZE"]),
                          "fetch_bytes_corrected_x2": fetch, "write_bytes": write}
    print(key, fetch + write)
json.dump(d, open(out, "w"), indent=1)
