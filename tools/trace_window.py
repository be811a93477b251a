#!/usr/bin/env python3
"""Dev tool: the last N kernels of a rocprofv3 --kernel-trace (csv) as a timeline: start / end relative to the first of them,
queue, name.   python tools/trace_window.py <dir> [N]"""
import csv
import glob
import sys

rows = []
for f in glob.glob(sys.argv[1] + "/**/*kernel_trace.csv", recursive=True):
    rows += list(csv.DictReader(open(f)))
n = int(sys.argv[2]) if len(sys.argv) > 2 else 12
rows.sort(key=lambda r: int(r["Start_Timestamp"]))
rows = rows[-n:]
t0 = int(rows[0]["Start_Timestamp"])
for r in rows:
    print("%9.1f %9.1f us  q%-3s %s" % ((int(r["Start_Timestamp"]) - t0) / 1e3, (int(r["End_Timestamp"]) - t0) / 1e3,
                                      r.get("Queue_Id", "?"), r["Kernel_Name"].split("(")[0][-48:]))
