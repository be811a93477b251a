#!/usr/bin/env python3
"""HBM traffic of every Pippenger kernel from the rocprofv3 PMC counters, collected as MI355X_MICROARCH.md (section HBM /
"rocprofv3 PMC slots") prescribes: FETCH_SIZE and WRITE_SIZE in SEPARATE passes (FETCH_SIZE costs 3 of the 4 TCC slots,
WRITE_SIZE 2), each pass with --kernel-trace only; the counters report KiB per dispatch; on gfx950 FETCH_SIZE tallies
128-byte requests at 64 B, so fetched bytes = 2 x FETCH_SIZE (Infinity-Cache hits are included: it is fabric-side
traffic, an upper bound on HBM bytes); WRITE_SIZE is taken as counted.

Runs ON THE GPU BOX (needs rocprofv3 and a device):
    python tools/pmc_traffic.py [--tag r02] [--log2n 20]
writes profiles/<tag>_pmc_hbm_traffic.json (+ .txt table) stamped with the kernel-source hash; bench.py quotes
`roofline.traffic` from it only while that hash equals the tree's."""
import argparse
import collections
import csv
import glob
import importlib.util
import json
import os
import re
import shutil
import subprocess
import sys
import tempfile

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def one_pass(counter, log2n, out):
    cmd = ["rocprofv3", "--kernel-trace", "--pmc", counter, "--output-format", "csv", "-d", out, "--",
           sys.executable, os.path.join(ROOT, "bench.py"), "--steps", "3", "--warmup", "1", "--no-cpu-baseline",
           "--no-secondary", "--inflight", "1", "--log2n", str(log2n)]
    env = dict(os.environ, TMPDIR="/tmp")
    subprocess.run(cmd, check=True, cwd="/tmp", env=env, stdout=subprocess.DEVNULL, stderr=subprocess.DEVNULL)
    agg, calls, seen = collections.defaultdict(float), collections.Counter(), set()
    for f in glob.glob(os.path.join(out, "**", "*counter_collection.csv"), recursive=True):
        for r in csv.DictReader(open(f)):
            if r["Counter_Name"] != counter:
                continue
            m = re.search(r"(k_[a-z0-9_]+)", r["Kernel_Name"])
            name = m.group(1) if m else r["Kernel_Name"].split("(")[0][:40]
            agg[name] += float(r["Counter_Value"])
            if (name, r["Dispatch_Id"]) not in seen:
                seen.add((name, r["Dispatch_Id"]))
                calls[name] += 1
    return {k: (agg[k] / calls[k], calls[k]) for k in agg}


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--tag", default="r02")
    ap.add_argument("--log2n", type=int, default=20)
    ap.add_argument("--out-dir", default=os.path.join(ROOT, "profiles"),
                    help="gpurun only merges gpurun_out/ back: use --out-dir gpurun_out there and copy the files into profiles/")
    a = ap.parse_args()
    spec = importlib.util.spec_from_file_location("_h", os.path.join(ROOT, "snark-verifier_amd", "_srchash.py"))
    h = importlib.util.module_from_spec(spec)
    spec.loader.exec_module(h)
    tmp = tempfile.mkdtemp(prefix="pmc_")
    fetch = one_pass("FETCH_SIZE", a.log2n, os.path.join(tmp, "fetch"))
    write = one_pass("WRITE_SIZE", a.log2n, os.path.join(tmp, "write"))
    shutil.rmtree(tmp, ignore_errors=True)
    kernels = {}
    for k in sorted(set(fetch) | set(write)):
        f, cf = fetch.get(k, (0.0, 0))
        w, cw = write.get(k, (0.0, 0))
        kernels[k] = {"calls": max(cf, cw), "fetch_size_kib": f, "write_size_kib": w,
                      "bytes_as_counted": (f + w) * 1024.0, "bytes_corrected": (2.0 * f + w) * 1024.0}
    rec = {
        "kernel_source_hash": h.kernel_source_hash(), "log2n": a.log2n,
        "command": "rocprofv3 --kernel-trace --pmc {FETCH_SIZE | WRITE_SIZE} (two passes) -- python bench.py --steps 3 --warmup 1 "
                   "--no-cpu-baseline --no-secondary --inflight 1 --log2n %d" % a.log2n,
        "unit": "KiB per dispatch as reported; bytes_corrected = (2 x FETCH_SIZE + WRITE_SIZE) x 1024 "
                "(gfx950: FETCH_SIZE counts 128-byte requests as 64 B; Infinity-Cache hits included)",
        "kernels": kernels,
    }
    base = os.path.join(a.out_dir, "%s_pmc_hbm_traffic%s" % (a.tag, "" if a.log2n == 20 else "_2p%d" % a.log2n))
    with open(base + ".json", "w") as f:
        json.dump(rec, f, indent=1)
    with open(base + ".txt", "w") as f:
        f.write("# %s\n# %s\n# kernel_source_hash %s\n" % (rec["command"], rec["unit"], rec["kernel_source_hash"]))
        f.write("%-28s %6s %14s %14s %18s\n" % ("kernel", "calls", "FETCH_KiB", "WRITE_KiB", "corrected_MiB"))
        for k, v in sorted(kernels.items(), key=lambda kv: -kv[1]["bytes_corrected"]):
            f.write("%-28s %6d %14.1f %14.1f %18.1f\n" % (k, v["calls"], v["fetch_size_kib"], v["write_size_kib"], v["bytes_corrected"] / 2**20))
    print(open(base + ".txt").read())


if __name__ == "__main__":
    main()
