"""A/B of the pipelined Poseidon aggregation job (host/aggregation.hpp `aggregate_pipelined`) on the GPU box:
1 024 distinct proofs (tests/golden/bench_plonk_gwc19_poseidon_1024.bin) through `snarkv_host_aggregate`, per transcript
route and pipeline setting, median / min / p95 of CALLS individually timed calls (the library's own wall clock), with
the phase split of the median call.  Interleaved rounds, so that clock drift hits every variant alike.
    python tools/ab_pipeline.py [--calls 25] [--rounds 3] [--chunks 64,128,256] [--sizes 1024,256,512]"""
import argparse
import os
import sys
import time

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)


def throttled():
    try:
        for ln in open("/sys/fs/cgroup/cpu.stat"):
            if ln.startswith("nr_throttled"):
                return int(ln.split()[1])
    except OSError:
        pass
    return 0


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--calls", type=int, default=25)
    ap.add_argument("--rounds", type=int, default=3)
    ap.add_argument("--chunks", default="64,128,256")
    ap.add_argument("--dthreads", default="1,2")
    ap.add_argument("--sizes", default="1024")
    ap.add_argument("--fixture", default="bench_plonk_gwc19_poseidon_1024.bin")
    ap.add_argument("--mos", type=int, default=0)
    ap.add_argument("--pace", type=float, default=0.0, help="milliseconds of idle time after every call")
    ap.add_argument("--slowest", type=int, default=0, help="also print the phases of the N slowest calls of every variant")
    args = ap.parse_args()
    from snark_verifier_amd import host_api as H

    fx = H.read_fixture(os.path.join(ROOT, "tests", "golden", args.fixture))
    hp, hdk = H.Protocol(fx["protocol"]), H.DecidingKey(fx["dk"])
    threads = max(1, min(64, os.cpu_count() or 1))
    # prefixes of the batch: byte offsets of proof m / instance set m
    poffs, off = [0], 0
    for _ in range(fx["n"]):
        off += 4 + int.from_bytes(fx["proofs"][off:off + 4], "little")
        poffs.append(off)
    ioffs, off = [0], 0
    inst = fx["instances"]
    for _ in range(fx["n"]):
        cols = int.from_bytes(inst[off:off + 4], "little")
        off += 4
        for _c in range(cols):
            off += 4 + 32 * int.from_bytes(inst[off:off + 4], "little")
        ioffs.append(off)
    # (name, transcript route, environment of the call; None = unset)
    base = {"SNARKV_HOST_PIPELINE_MIN": None, "SNARKV_HOST_PIPELINE_CHUNK": None, "SNARKV_HOST_PIPELINE_DEVICE_THREADS": None}
    variants = [("host route, pipeline off", 1, dict(base, SNARKV_HOST_PIPELINE_MIN="0")),
                ("device route (no pipeline exists)", 2, dict(base, SNARKV_HOST_PIPELINE_MIN="0"))]
    for c in args.chunks.split(","):
        for d in args.dthreads.split(","):
            variants.append(("host route, pipelined, chunk %s, %s device thread(s)" % (c, d), 1,
                             dict(base, SNARKV_HOST_PIPELINE_MIN="2", SNARKV_HOST_PIPELINE_CHUNK=c, SNARKV_HOST_PIPELINE_DEVICE_THREADS=d)))
    variants.append(("auto route (as shipped)", 3, dict(base)))
    for m in [int(x) for x in args.sizes.split(",")]:
        m = min(m, fx["n"])
        pb, ib = fx["proofs"][:poffs[m]], inst[:ioffs[m]]
        res = {v[0]: [] for v in variants}
        cpu = {v[0]: 0.0 for v in variants}
        thr = {v[0]: 0 for v in variants}
        ref = None
        for rnd in range(args.rounds):
            for name, tk, env in variants:
                for k, v in env.items():
                    if v is None:
                        os.environ.pop(k, None)
                    else:
                        os.environ[k] = v
                for _ in range(3):
                    H.aggregate(hp, hdk, ib, pb, m, args.mos, tk, threads)
                thr0, c0 = throttled(), time.process_time()
                for _ in range(args.calls):
                    ok, acc, tm = H.aggregate(hp, hdk, ib, pb, m, args.mos, tk, threads, timings=True)
                    assert ok
                    ref = ref or acc
                    assert acc == ref, name
                    res[name].append(tm)
                    if args.pace:  # keep the cgroup's CPU quota out of the measurement: idle until the budget has caught up
                        time.sleep(args.pace * 1e-3)
                cpu[name] += (time.process_time() - c0) / args.calls * 1e3
                thr[name] += throttled() - thr0
        print("## %d proofs of %s, %d host threads, %d rounds x %d calls per variant; every accumulator identical"
              % (m, args.fixture, threads, args.rounds, args.calls))
        for name, _, _ in variants:
            r = sorted(res[name], key=lambda t: t["total"])
            med = r[len(r) // 2]
            print("%-58s median %7.3f  min %7.3f  p95 %7.3f ms | of the median call: read %.2f algebra %.2f msm %.2f accumulate %.2f decide %.2f"
                  % (name, med["total"], r[0]["total"], r[int(len(r) * 0.95)]["total"], med["read_proofs"], med["fr_algebra"],
                     med["msm_device"], med["accumulate"], med["decide"]))
            print("      CPU time per call %.0f ms (all threads, process clock); cgroup throttle events during this variant: %d"
                  % (cpu[name] / args.rounds, thr[name]))
            if args.slowest:
                for t in r[-args.slowest:]:
                    print("      slow call: total %7.3f | read %.2f algebra %.2f msm %.2f accumulate %.2f decide %.2f"
                          % (t["total"], t["read_proofs"], t["fr_algebra"], t["msm_device"], t["accumulate"], t["decide"]))
        sys.stdout.flush()


if __name__ == "__main__":
    main()
