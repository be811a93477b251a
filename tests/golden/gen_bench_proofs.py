"""Generates tests/golden/bench_plonk_{gwc19_evm,gwc19_poseidon,bdfg21_poseidon}_64.bin: the C3 workload of
BASELINE.json as REAL INPUT BYTES -- one StandardPlonk-shaped protocol, 64
instance sets and 64 proofs forged under the toy SRS (oracle/plonk.py), Keccak
or Poseidon transcript for the proofs, GWC19 multi-open -- so that bench.py can time the verifier end to
end (proof bytes in, accept out) without importing the oracle.
Layout: magic 'SVB1' | u32 n | u32 plen | protocol | u32 ilen | instances | u32 prlen | proofs (u32 len || bytes each)
        | dk (64 + 128 + 128) | expected aggregated accumulator (128)
Run from the repo root:  python tests/golden/gen_bench_proofs.py

The accumulation step's transcript is of the proofs' family (Keccak for Keccak proofs; Poseidon for Poseidon proofs, as
examples/evm-verifier-with-accumulator.rs:361,375).  Every file carries the trailer described below.

`python tests/golden/gen_bench_proofs.py 1024` writes the C5 workloads (BASELINE.json configs[4],
reference call pattern snark-verifier/examples/evm-verifier-with-accumulator.rs:357-385 x1024):
bench_plonk_gwc19_{evm,poseidon}_1024.bin, 1024 DISTINCT instance sets and proofs (per-proof seeded generators, forged in
a process pool), same layout plus a trailer  u32 n | n x 128 B  = every proof's own accumulator as the
oracle's PlonkSuccinctVerifier computed it (decide_all over 1024 distinct accumulators, per-proof parity)."""
import os
import random
import struct
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, os.path.join(ROOT, "oracle"))
sys.path.insert(0, os.path.join(ROOT, "tests"))
import bn254 as O
import kzg as K
import plonk as P
import plonk_synth as S
import transcript as T

SECRET = 0x1F2E3D4C5B6A79887766554433221100AABBCCDDEEFF


def _accumulate(accs, kind):
    """KzgAs::create_proof (non-zk) over the accumulators with a fresh transcript of the proofs' family
    (accumulation.rs:148-197; the reference's example uses Poseidon for both, evm-verifier-with-accumulator.rs:361,375)"""
    t = T.EvmTranscript() if kind == "evm" else T.PoseidonTranscript()
    for lhs, rhs in accs:
        t.common_ec_point(lhs)
        t.common_ec_point(rhs)
    r = t.squeeze_challenge()
    return K.kzg_as_verify(accs, r)


def main(n=64, kind="evm", mos="gwc19"):
    rng = random.Random((0xBE2C if kind == "evm" else 0xBE2D) + (0 if mos == "gwc19" else 0x100))
    TR = T.EvmTranscript if kind == "evm" else T.PoseidonTranscript
    pr, dl = S.standard_plonk_protocol(rng)
    insts, proofs, accs = [], [], []
    for i in range(n):
        inst = [[rng.randrange(O.R) for _ in range(m)] for m in pr["num_instance"]]
        proof = P.forge_proof(pr, inst, SECRET, lambda: TR(), mos, rng, dl)
        pf = P.plonk_proof_read(pr, inst, TR(proof), mos)
        accs += P.succinct_verify(O.G1_GEN, pr, inst, pf, mos)
        insts.append(inst)
        proofs.append(proof)
    agg = _accumulate(accs, kind)
    assert agg[0] == O.g1_mul(agg[1], SECRET)
    pb = S.pack_protocol(pr)
    ib = b"".join(S.pack_instances(i) for i in insts)
    prb = b"".join(struct.pack("<I", len(p)) + p for p in proofs)
    dk = O.g1_to_bytes(O.G1_GEN) + O.g2_to_bytes(O.G2_GEN) + O.g2_to_bytes(O.g2_mul(O.G2_GEN, SECRET))
    blob = (b"SVB1" + struct.pack("<I", n) + struct.pack("<I", len(pb)) + pb + struct.pack("<I", len(ib)) + ib
            + struct.pack("<I", len(prb)) + prb + dk + O.g1_to_bytes(agg[0]) + O.g1_to_bytes(agg[1])
            + struct.pack("<I", n) + b"".join(O.g1_to_bytes(l) + O.g1_to_bytes(rh) for l, rh in accs))
    path = os.path.join(ROOT, "tests", "golden", "bench_plonk_%s_%s_%d.bin" % (mos, kind, n))
    with open(path, "wb") as f:
        f.write(blob)
    print("wrote", path, len(blob), "bytes")


def _forge_one(args):
    seed, i, kind = args
    rng = random.Random(seed)
    pr, dl = S.standard_plonk_protocol(rng)          # the same protocol in every worker (same seed)
    rng = random.Random((seed << 20) ^ (i + 1))      # per-proof stream: distinct instances, blinds, evaluations
    TR = T.EvmTranscript if kind == "evm" else T.PoseidonTranscript
    inst = [[rng.randrange(O.R) for _ in range(m)] for m in pr["num_instance"]]
    proof = P.forge_proof(pr, inst, SECRET, lambda: TR(), "gwc19", rng, dl)
    pf = P.plonk_proof_read(pr, inst, TR(proof), "gwc19")
    accs = P.succinct_verify(O.G1_GEN, pr, inst, pf, "gwc19")
    assert len(accs) == 1
    return inst, proof, accs[0]


def main_distinct(n=1024, kind="evm", seed=0xC5C5):
    import multiprocessing as mp

    pr, _ = S.standard_plonk_protocol(random.Random(seed))
    with mp.Pool() as pool:
        res = pool.map(_forge_one, [(seed, i, kind) for i in range(n)], chunksize=8)
    insts, proofs, accs = [r[0] for r in res], [r[1] for r in res], [r[2] for r in res]
    assert len(set(proofs)) == n
    agg = _accumulate(accs, kind)
    assert agg[0] == O.g1_mul(agg[1], SECRET)
    pb = S.pack_protocol(pr)
    ib = b"".join(S.pack_instances(i) for i in insts)
    prb = b"".join(struct.pack("<I", len(p)) + p for p in proofs)
    dk = O.g1_to_bytes(O.G1_GEN) + O.g2_to_bytes(O.G2_GEN) + O.g2_to_bytes(O.g2_mul(O.G2_GEN, SECRET))
    blob = (b"SVB1" + struct.pack("<I", n) + struct.pack("<I", len(pb)) + pb + struct.pack("<I", len(ib)) + ib
            + struct.pack("<I", len(prb)) + prb + dk + O.g1_to_bytes(agg[0]) + O.g1_to_bytes(agg[1])
            + struct.pack("<I", n) + b"".join(O.g1_to_bytes(l) + O.g1_to_bytes(rh) for l, rh in accs))
    path = os.path.join(ROOT, "tests", "golden", "bench_plonk_gwc19_%s_%d.bin" % (kind, n))
    with open(path, "wb") as f:
        f.write(blob)
    print("wrote", path, len(blob), "bytes")


if __name__ == "__main__":
    if len(sys.argv) > 1:  # `... 1024 [evm|poseidon]`: the C5 workloads (both kinds without the second argument)
        for kind in (sys.argv[2:] or ["evm", "poseidon"]):
            main_distinct(int(sys.argv[1]), kind)
    else:
        main(kind="evm")
        main(kind="poseidon")
        main(kind="poseidon", mos="bdfg21")  # the SDK's default SHPLONK shape (snark-verifier-sdk/src/lib.rs:41)
