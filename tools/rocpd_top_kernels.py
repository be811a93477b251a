"""Dev tool: export the `top_kernels` view of a rocprofv3 rocpd database as the
Name,Calls,TotalDurationUs,AverageUs,Percentage CSV kept under profiles/.
  python tools/rocpd_top_kernels.py <results.db> <out.csv>"""
import csv
import sqlite3
import sys

db = sqlite3.connect(sys.argv[1])
cols = [r[1] for r in db.execute("pragma table_info(top_kernels)")]
rows = db.execute("select * from top_kernels").fetchall()
pick = {c.lower(): i for i, c in enumerate(cols)}


def col(row, *names):
    for n in names:
        if n in pick:
            return row[pick[n]]
    raise KeyError(names, cols)


with open(sys.argv[2], "w", newline="") as f:
    w = csv.writer(f, quoting=csv.QUOTE_NONNUMERIC)
    f.write("Name,Calls,TotalDurationUs,AverageUs,Percentage\n")
    for r in rows:
        # the view reports microseconds (rocprofv3 7.x: `total_duration`, `average`)
        w.writerow([col(r, "name"), int(col(r, "total_calls", "calls")), round(col(r, "total_duration"), 3),
                    round(col(r, "average"), 3), round(col(r, "percentage", "percent"), 4)])
