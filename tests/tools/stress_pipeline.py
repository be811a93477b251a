"""Stress of the pipelined Poseidon aggregation job (host/aggregation.hpp aggregate_pipelined) on the GPU box: for `secs`
seconds, random sub-batches of the 1 024-proof fixture with random chunk sizes / device-thread counts / first-chunk ramp,
a third of them with one to three corrupted proofs at random places (a flipped evaluation bit, a broken point encoding, a
scalar made non-canonical), each job run pipelined AND unpipelined: the two outcomes -- verdict + accumulator, or error code +
text -- must be identical every time; several jobs from several application threads at once every tenth round.  A hang
shows as the caller's timeout.
    python tests/tools/stress_pipeline.py [secs] [seed]"""
import os
import random
import sys
import threading
import time

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)


def main():
    secs = float(sys.argv[1]) if len(sys.argv) > 1 else 60
    rng = random.Random(int(sys.argv[2]) if len(sys.argv) > 2 else 1)
    from snark_verifier_amd import host_api as H

    fx = H.read_fixture(os.path.join(ROOT, "tests", "golden", "bench_plonk_gwc19_poseidon_1024.bin"))
    hp, hdk = H.Protocol(fx["protocol"]), H.DecidingKey(fx["dk"])
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

    def outcome(ib, pb, m, threads):
        try:
            ok, acc = H.aggregate(hp, hdk, ib, pb, m, H.MOS_GWC19, H.TRANSCRIPT_POSEIDON, threads)
            return ("ok", ok, acc)
        except H.HostError as e:
            return ("err", e.code, str(e))

    def setenv(pmin, chunk, dthreads, ramp):
        os.environ["SNARKV_HOST_PIPELINE_MIN"] = pmin
        os.environ["SNARKV_HOST_PIPELINE_CHUNK"] = str(chunk)
        os.environ["SNARKV_HOST_PIPELINE_DEVICE_THREADS"] = str(dthreads)
        if ramp:
            os.environ.pop("SNARKV_HOST_PIPELINE_NO_RAMP", None)
        else:
            os.environ["SNARKV_HOST_PIPELINE_NO_RAMP"] = "1"

    t0, rounds, bad_rounds, concurrent = time.time(), 0, 0, 0
    while time.time() - t0 < secs:
        lo = rng.randrange(0, fx["n"] - 2)
        m = rng.choice([2, 3, 9, 64, 129, 300, 511, 1024, rng.randrange(2, 600)])
        m = min(m, fx["n"] - lo)
        pb = bytearray(fx["proofs"][poffs[lo]:poffs[lo + m]])
        ib = inst[ioffs[lo]:ioffs[lo + m]]
        if rng.random() < 0.33:
            bad_rounds += 1
            for _ in range(rng.randrange(1, 4)):
                j = rng.randrange(m)
                o, ln = poffs[lo + j] - poffs[lo] + 4, poffs[lo + j + 1] - poffs[lo + j] - 4
                kind = rng.randrange(3)
                if kind == 0:
                    pb[o + ln - 1 - rng.randrange(32 * 6)] ^= 1 << rng.randrange(8)
                elif kind == 1:
                    pb[o + 32 * rng.randrange(3) + 31] ^= rng.choice([0x3F, 0x80, 0x40])
                else:
                    for k in range(32):
                        pb[o + ln - 64 + k] = 0xFF
        pb = bytes(pb)
        threads = rng.choice([64, 64, 16, 3])
        chunk, dthreads, ramp = rng.choice([1, 7, 32, 64, 100, 128, 256, 5000]), rng.choice([1, 2, 3]), rng.random() < 0.7
        setenv("0", chunk, dthreads, ramp)
        ref = outcome(ib, pb, m, threads)
        setenv("2", chunk, dthreads, ramp)
        if rounds % 10 == 9:  # several application threads at once, each with its own copy of the job
            concurrent += 1
            got = [None] * 4
            ts = [threading.Thread(target=lambda k=k: got.__setitem__(k, outcome(ib, pb, m, threads))) for k in range(4)]
            for t in ts:
                t.start()
            for t in ts:
                t.join()
        else:
            got = [outcome(ib, pb, m, threads)]
        for g in got:
            if g != ref:
                print("MISMATCH: lo %d m %d threads %d chunk %d dthreads %d ramp %s\n  unpipelined %r\n  pipelined   %r"
                      % (lo, m, threads, chunk, dthreads, ramp, ref[:2], g[:2]))
                sys.exit(1)
        rounds += 1
    print("pipeline stress ok: %d rounds (%d with corrupted proofs, %d with four jobs at once), %.0f s" % (rounds, bad_rounds, concurrent, time.time() - t0))


if __name__ == "__main__":
    main()
