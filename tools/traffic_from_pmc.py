#!/usr/bin/env python3
"""profiles/traffic.json from a tools/pmc.sh run: HBM bytes per launch of the hot-path kernels, from the
FETCH_SIZE / WRITE_SIZE passes (separate --pmc runs).

Corrections per MI355X_MICROARCH.md (HBM section): both counters are in KiB; on gfx950 FETCH_SIZE reports
exactly half of the bytes of a wide (16 B/lane) coalesced streaming read -> doubled; WRITE_SIZE is exact for
streaming stores.  That x2 is a rule for STREAMS whose L2 misses are 128-byte requests tallied at 64 bytes; a
gather of 64-byte pixels (the fp16 WarpV2: 4 lanes x 16 B per corner) issues genuine 64-byte requests and the x2
over-counts it (round 3: 1.92 x the compulsory bytes; uncorrected 0.96 x).  Round 4: where the pass with the
request-size counters exists (tools/pmc.sh pass H: TCC_EA0_RDREQ_32B / _64B / _128B), the read bytes are
32 n32 + 64 n64 + 128 n128 -- no per-kernel rule at all -- and every key records which rule produced its number
(`read_rule`) with the other figure beside it; writes likewise from pass I (64 n64 + 32 (n - n64)) when present.  The file is stamped with the sha256 of the kernel sources (qpwcnet_amd._hip.source_sha256):
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
     "note": "separate rocprofv3 --pmc passes over tools/cv84_launch.py (tools/make_traffic.sh): read bytes from "
             "the L2's memory-side request counters by request size (32 / 64 / 128 B; pass H) where collected -- "
             "each key's `read_rule` says which rule made its number, with FETCH_SIZE x1 / x2 beside it -- "
             "writes from WRITE_SIZE KiB; the cost-volume launches write 84-float pixels (81 channels + 3 zeroed "
             "pads), the algorithmic bytes count 81"}
d.update({k: v for k, v in prev.items() if k not in d})
for key, sub in pairs:
    vals = {"FETCH_SIZE": [], "WRITE_SIZE": [], "TCC_EA0_RDREQ_sum": [], "TCC_EA0_RDREQ_32B_sum": [],
            "TCC_EA0_RDREQ_64B_sum": [], "TCC_EA0_RDREQ_128B_sum": [], "TCC_EA0_WRREQ_sum": [],
            "TCC_EA0_WRREQ_64B_sum": [], "TCC_EA0_RDREQ_DRAM_sum": [], "TCC_EA0_WRREQ_DRAM_sum": []}
    for r in rows:
        if any(a in r["Kernel_Name"] for a in sub.split("|")) and r["Counter_Name"] in vals:
            vals[r["Counter_Name"]].append(float(r["Counter_Value"]))
    if not vals["FETCH_SIZE"] or not vals["WRITE_SIZE"]:
        print("no counters for", key, sub)
        continue
    mean = lambda k: sum(vals[k]) / len(vals[k]) if vals[k] else None   # noqa: E731
    fetch_x2 = mean("FETCH_SIZE") * 1024 * 2
    write = mean("WRITE_SIZE") * 1024
    det = {"kernel": sub, "launches": len(vals["FETCH_SIZE"]), "FETCH_SIZE_KiB_raw": mean("FETCH_SIZE"),
           "fetch_bytes_FETCH_SIZE_x2": fetch_x2, "fetch_bytes_FETCH_SIZE_x1": fetch_x2 / 2,
           "write_bytes_WRITE_SIZE": write}
    n32, n64, n128, nrd = (mean("TCC_EA0_RDREQ_32B_sum"), mean("TCC_EA0_RDREQ_64B_sum"),
                           mean("TCC_EA0_RDREQ_128B_sum"), mean("TCC_EA0_RDREQ_sum"))
    if None not in (n32, n64, n128, nrd):
        # do the three sizes partition the requests?  (if 64-B requests are "the rest", say so)
        rest = nrd - n32 - n128
        n64_used = n64 if abs(n32 + n64 + n128 - nrd) <= 0.02 * max(nrd, 1.0) else rest
        fetch = 32 * n32 + 64 * n64_used + 128 * n128
        det.update({"read_requests": nrd, "read_requests_32B": n32, "read_requests_64B": n64,
                    "read_requests_128B": n128, "read_requests_64B_used": n64_used,
                    "fetch_bytes_by_request_size": fetch,
                    "read_rule": "request sizes (32 n32 + 64 n64 + 128 n128, TCC_EA0_RDREQ_*; pass H)"})
    else:
        fetch = fetch_x2
        det["read_rule"] = "FETCH_SIZE KiB x 2 (the guide's rule for 16-B/lane streams; request-size pass not run)"
    nwr, nw64 = mean("TCC_EA0_WRREQ_sum"), mean("TCC_EA0_WRREQ_64B_sum")
    if None not in (nwr, nw64):
        det.update({"write_requests": nwr, "write_requests_64B": nw64,
                    "write_bytes_by_request_size": 64 * nw64 + 32 * (nwr - nw64)})
    for k in ("TCC_EA0_RDREQ_DRAM_sum", "TCC_EA0_WRREQ_DRAM_sum"):
        if mean(k) is not None:
            det[k] = mean(k)
    det["fetch_bytes"] = fetch
    d[key] = fetch + write
    d[key + "_detail"] = det
    print(key, fetch + write, det["read_rule"])
json.dump(d, open(out, "w"), indent=1)
