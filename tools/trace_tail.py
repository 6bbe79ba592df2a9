#!/usr/bin/env python3
"""The last launches of the overlap kernels in a rocprofv3 kernel trace: start (relative), duration, grid -- how the two streams of a
pipelined fmd_ovlp_dev call interleave.  Usage: trace_tail.py kernel_trace.csv [n=90]"""
import csv, sys
rows = [r for r in csv.DictReader(open(sys.argv[1])) if "k_ovl" in r["Kernel_Name"]]
n = int(sys.argv[2]) if len(sys.argv) > 2 else 90
rows.sort(key=lambda r: int(r["Start_Timestamp"]))
rows = rows[-n:]
t0 = int(rows[0]["Start_Timestamp"]) if rows else 0
for r in rows:
    s, e = int(r["Start_Timestamp"]), int(r["End_Timestamp"])
    name = r["Kernel_Name"].split("(")[0].replace("void ", "")
    print("%-34s start %9.3f ms  end %9.3f ms  dur %8.3f ms  grid %s" % (name[:34], (s - t0) / 1e6, (e - t0) / 1e6, (e - s) / 1e6, r.get("Grid_Size", "")))
