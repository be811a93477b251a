#!/usr/bin/env python3
"""Writes the hash sidecar of a tracked rocprofv3 kernel-stats CSV: per kernel (short name, template arguments dropped)
the call count and average duration, with the kernel-source hash of THIS tree -- bench.py quotes `kernel_us_profile` from
it only while the hash matches the sources it runs.
  python tools/stats_sidecar.py profiles/r05_rocprofv3_kernel_stats_sequential.csv  ->  ..._kernel_stats_sequential.json"""
import csv
import importlib.util
import json
import os
import re
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
spec = importlib.util.spec_from_file_location("_srchash", os.path.join(ROOT, "snark-verifier_amd", "_srchash.py"))
m = importlib.util.module_from_spec(spec)
spec.loader.exec_module(m)

src = sys.argv[1]
kernels = {}
for row in csv.DictReader(open(src)):
    name = re.sub(r"^void\s+", "", row["Name"]).split("(")[0]
    short = re.sub(r"<.*$", "", name.split("::")[-1])
    k = kernels.setdefault(short, {"calls": 0, "total_us": 0.0, "variants": []})
    k["calls"] += int(row["Calls"])
    k["total_us"] += float(row["TotalDurationUs"])
    k["variants"].append(name)
for k in kernels.values():
    k["avg_us"] = k["total_us"] / max(1, k["calls"])
out = re.sub(r"rocprofv3_kernel_stats_(\w+)\.csv$", r"kernel_stats_\1.json", src)
assert out != src, "expected a *_rocprofv3_kernel_stats_<name>.csv"
json.dump({"kernel_source_hash": m.kernel_source_hash(), "csv": os.path.basename(src), "kernels": kernels},
          open(out, "w"), indent=1, sort_keys=True)
print("wrote", out)
