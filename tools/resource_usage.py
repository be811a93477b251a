#!/usr/bin/env python3
"""Kernel resource usage of every HIP kernel (hipcc -Rpass-analysis=kernel-resource-usage, the build's flags) as one table
-> profiles/<tag>_kernel_resource_usage.txt.  python tools/resource_usage.py [--tag r02]"""
import importlib.util, os, re, subprocess, sys, tempfile
from concurrent.futures import ThreadPoolExecutor

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
spec = importlib.util.spec_from_file_location("_b", os.path.join(ROOT, "snark-verifier_amd", "build.py"))
b = importlib.util.module_from_spec(spec); spec.loader.exec_module(b)
tag = sys.argv[sys.argv.index("--tag") + 1] if "--tag" in sys.argv else "r02"


def one(unit):
    with tempfile.TemporaryDirectory() as d:
        r = subprocess.run([b.HIPCC] + b.FLAGS + ["-c", os.path.join(b.CSRC, unit + ".hip"), "-o", os.path.join(d, "x.o"),
                            "-Rpass-analysis=kernel-resource-usage"], capture_output=True, text=True)
    rows, cur = [], None
    for line in r.stderr.splitlines():
        m = re.search(r"remark: Function Name: (\S+)", line)
        if m:
            name = subprocess.run(["c++filt", m.group(1)], capture_output=True, text=True).stdout.strip().split("(")[0]
            cur = {"kernel": name.replace("snarkv::", ""), "unit": unit}
            rows.append(cur)
            continue
        m = re.search(r"remark:\s+(TotalSGPRs|VGPRs|AGPRs|ScratchSize \[bytes/lane\]|Occupancy \[waves/SIMD\]|LDS Size \[bytes/block\]|VGPRs Spill): (\d+)", line)
        if m and cur is not None:
            cur[m.group(1).split(" [")[0]] = int(m.group(2))
    return rows


with ThreadPoolExecutor(max_workers=8) as ex:
    allrows = [r for rows in ex.map(one, b.UNITS) for r in rows]
out = os.path.join(ROOT, "profiles", "%s_kernel_resource_usage.txt" % tag)
with open(out, "w") as f:
    f.write("# hipcc %s -Rpass-analysis=kernel-resource-usage (gfx950), every __global__ kernel of libsnarkv_amd.so\n" % " ".join(b.FLAGS))
    f.write("%-14s %-34s %6s %6s %8s %7s %6s %9s\n" % ("unit", "kernel", "VGPRs", "SGPRs", "scratchB", "spills", "waves", "LDS_B"))
    for r in allrows:
        f.write("%-14s %-34s %6d %6d %8d %7d %6d %9d\n" % (r["unit"], r["kernel"][:34], r.get("VGPRs", 0), r.get("TotalSGPRs", 0),
                r.get("ScratchSize", 0), r.get("VGPRs Spill", 0), r.get("Occupancy", 0), r.get("LDS Size", 0)))
print(open(out).read())
