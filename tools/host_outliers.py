#!/usr/bin/env python3
"""Tail latency of the end-to-end aggregation: N calls of snarkv_host_aggregate on one fixture, every
call's phase split from the library's own clocks; prints the distribution per phase and every call above 1.5 x the median
total with ITS phases -- which phase stalls is what an allocator / pool fix has to explain.
  python tools/host_outliers.py [--fixture bench_plonk_gwc19_evm_1024.bin] [--rep 1] [--kind 0] [--calls 100] [--threads 64]"""
import argparse
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
os.environ.setdefault("GPU_MAX_HW_QUEUES", "16")


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--fixture", default="bench_plonk_gwc19_evm_1024.bin")
    ap.add_argument("--rep", type=int, default=1)
    ap.add_argument("--kind", type=int, default=0)
    ap.add_argument("--mos", type=int, default=0)
    ap.add_argument("--calls", type=int, default=100)
    ap.add_argument("--threads", type=int, default=64)
    ap.add_argument("--label", default="")
    a = ap.parse_args()
    from snark_verifier_amd import host_api as H

    fx = H.read_fixture(os.path.join(ROOT, "tests", "golden", a.fixture))
    hp, hdk = H.Protocol(fx["protocol"]), H.DecidingKey(fx["dk"])
    inst, prf, n = fx["instances"] * a.rep, fx["proofs"] * a.rep, fx["n"] * a.rep
    for _ in range(3):
        H.aggregate(hp, hdk, inst, prf, n, a.mos, a.kind, a.threads, timings=True)
    recs = []
    for _ in range(a.calls):
        ok, acc, tm = H.aggregate(hp, hdk, inst, prf, n, a.mos, a.kind, a.threads, timings=True)
        assert ok
        recs.append(tm)
    names = ("read_proofs", "fr_algebra", "msm_device", "accumulate", "decide", "total")

    def q(xs, f):
        xs = sorted(xs)
        return xs[min(len(xs) - 1, int(f * len(xs)))]

    med = q([r["total"] for r in recs], 0.5)
    out = [r for r in recs if r["total"] > 1.5 * med]
    print("%s %s x%d kind %d, %d proofs, %d threads, %d calls: outliers (> 1.5 x median total) %d" %
          (a.label, a.fixture, a.rep, a.kind, n, a.threads, a.calls, len(out)))
    for k in names:
        xs = [r[k] for r in recs]
        print("   %-12s min %.3f  median %.3f  p95 %.3f  max %.3f" % (k, min(xs), q(xs, 0.5), q(xs, 0.95), max(xs)))
    for r in out:
        print("   outlier: " + "  ".join("%s %.3f" % (k, r[k]) for k in names))


if __name__ == "__main__":
    main()
