"""SQ counters of a command's kernels, per kernel (separate rocprofv3 --pmc run, --kernel-trace only): VALU / SALU
instructions, wavefronts, wavefront-cycles, busy cycles -> instructions per wavefront and wavefront-cycles per issued VALU
instruction (what prices a latency chain: ~6 cycles when a wavefront has its SIMD to itself).
  python tools/pmc_sq.py [--out FILE] -- python tools/aggregate_job.py --proofs 64"""
import collections
import csv
import glob
import os
import re
import shutil
import subprocess
import sys
import tempfile

COUNTERS = ["SQ_INSTS_VALU", "SQ_INSTS_SALU", "SQ_WAVES", "SQ_WAVE_CYCLES", "SQ_BUSY_CYCLES", "SQ_ACTIVE_INST_VALU"]


def main():
    argv = sys.argv[1:]
    out_file = None
    if argv and argv[0] == "--out":
        out_file, argv = argv[1], argv[2:]
    if argv and argv[0] == "--":
        argv = argv[1:]
    tmp = tempfile.mkdtemp(prefix="pmcsq_")
    cmd = ["rocprofv3", "--kernel-trace", "--pmc"] + COUNTERS + ["--output-format", "csv", "-d", tmp, "--"] + argv
    subprocess.run(cmd, check=True, env=dict(os.environ, TMPDIR="/tmp"), stdout=subprocess.DEVNULL, stderr=subprocess.DEVNULL)
    agg = collections.defaultdict(lambda: collections.defaultdict(float))
    calls = collections.defaultdict(set)
    for f in glob.glob(os.path.join(tmp, "**", "*counter_collection.csv"), recursive=True):
        for r in csv.DictReader(open(f)):
            m = re.search(r"(k_[a-z0-9_]+)", r["Kernel_Name"])
            name = m.group(1) if m else r["Kernel_Name"].split("(")[0][:30]
            agg[name][r["Counter_Name"]] += float(r["Counter_Value"])
            calls[name].add(r["Dispatch_Id"])
    shutil.rmtree(tmp, ignore_errors=True)
    lines = ["# rocprofv3 --kernel-trace --pmc %s -- %s" % (" ".join(COUNTERS), " ".join(argv)),
             "# per launch (average over the calls); insts/wave = SQ_INSTS_VALU / SQ_WAVES, cyc/inst = SQ_WAVE_CYCLES / SQ_INSTS_VALU",
             "%-28s %6s %12s %12s %10s %14s %12s %12s %9s" % ("kernel", "calls", "INSTS_VALU", "INSTS_SALU", "WAVES", "WAVE_CYCLES", "BUSY_CYCLES",
                                                                "insts/wave", "cyc/inst")]
    for k in sorted(agg, key=lambda k: -agg[k]["SQ_WAVE_CYCLES"]):
        n = max(1, len(calls[k]))
        v = {c: agg[k][c] / n for c in COUNTERS}
        lines.append("%-28s %6d %12.4g %12.4g %10.4g %14.4g %12.4g %12.0f %9.2f" % (
            k[:28], n, v["SQ_INSTS_VALU"], v["SQ_INSTS_SALU"], v["SQ_WAVES"], v["SQ_WAVE_CYCLES"], v["SQ_BUSY_CYCLES"],
            v["SQ_INSTS_VALU"] / max(v["SQ_WAVES"], 1), v["SQ_WAVE_CYCLES"] / max(v["SQ_INSTS_VALU"], 1)))
    text = "\n".join(lines) + "\n"
    if out_file:
        open(out_file, "a").write(text)
    print(text)


if __name__ == "__main__":
    main()
