#!/usr/bin/env python3
"""Condense a rocprofv3 --kernel-trace CSV into a small timeline of its last N kernels:
start_us,dur_us,queue,stream,grid,name (for reading idle gaps / the critical path of one forward).

    python tools/ktrace.py <dir with *kernel_trace.csv> out.csv [N]
"""
import csv
import glob
import os
import sys


def main():
    src, dst = sys.argv[1], sys.argv[2]
    n = int(sys.argv[3]) if len(sys.argv) > 3 else 1400
    files = glob.glob(os.path.join(src, "**", "*kernel_trace.csv"), recursive=True)
    if not files:
        raise SystemExit("no kernel_trace.csv under " + src)
    rows = list(csv.DictReader(open(sorted(files)[-1])))
    rows.sort(key=lambda r: int(r["Start_Timestamp"]))
    rows = rows[-n:]
    t0 = int(rows[0]["Start_Timestamp"])
    with open(dst, "w") as f:
        f.write("start_us,dur_us,queue,stream,grid,name\n")
        for r in rows:
            s, e = int(r["Start_Timestamp"]), int(r["End_Timestamp"])
            f.write("%.2f,%.2f,%s,%s,%s,%s\n" % ((s - t0) / 1e3, (e - s) / 1e3, r.get("Queue_Id", ""),
                                               r.get("Stream_Id", ""), r.get("Grid_Size_X", r.get("Grid_Size", "")),
                                               r["Kernel_Name"][:80].replace(",", ";")))
    print("wrote", dst, len(rows), "kernels; columns:", list(rows[0].keys()))


if __name__ == "__main__":
    main()
