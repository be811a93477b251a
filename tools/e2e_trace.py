"""End-to-end aggregation from proof bytes with the host mirror built with -DSNARKV_HOST_TRACE=1 (stage times of
read_proofs_device_hashed and KzgAs::verify on stderr), by transcript kind.
  g++ ... -DSNARKV_HOST_TRACE=1 -o snark-verifier_amd/libsnarkv_host_trace.so   (tools/build_host_trace.sh)
  SNARKV_HOST_LIB=snark-verifier_amd/libsnarkv_host_trace.so python tools/e2e_trace.py --kind 2 --rep 16"""
import argparse
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
os.environ.setdefault("GPU_MAX_HW_QUEUES", "16")


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--kind", type=int, default=2, help="0 evm, 1 poseidon on the host, 2 poseidon on the device, 3 auto")
    ap.add_argument("--rep", type=int, default=16, help="copies of the 64-proof fixture")
    ap.add_argument("--calls", type=int, default=6)
    ap.add_argument("--threads", type=int, default=64)
    a = ap.parse_args()
    from snark_verifier_amd import host_api as H

    path = os.path.join(ROOT, "tests", "golden", "bench_plonk_gwc19_%s_64.bin" % ("evm" if a.kind == 0 else "poseidon"))
    fx = H.read_fixture(path)
    hp, hdk = H.Protocol(fx["protocol"]), H.DecidingKey(fx["dk"])
    for c in range(a.calls):
        ok, acc, tm = H.aggregate(hp, hdk, fx["instances"] * a.rep, fx["proofs"] * a.rep, fx["n"] * a.rep, H.MOS_GWC19, a.kind,
                                  a.threads, timings=True)
        print("call %d ok=%s %s" % (c, ok, {k: round(v, 3) for k, v in tm.items()}), flush=True)
        sys.stderr.flush()


if __name__ == "__main__":
    main()
