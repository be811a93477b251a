"""GPU: the PLONK succinct verifier front half of the host mirror
(snark-verifier_amd/host/plonk.hpp; reference verifier/plonk.rs:58-147,
verifier/plonk/proof.rs:52-349, verifier/plonk/protocol.rs) end to end:
proof bytes -> transcript -> expression evaluation -> MSMs on the MI355X ->
pairing decide, against oracle/plonk.py on proofs FORGED under a toy SRS
(no halo2 prover exists here; see oracle/plonk.py)."""
import ctypes
import os
import random

import pytest

import bn254 as O
import kzg as K
import plonk as P
import plonk_synth as S
import transcript as T
from hostfmt import g1, load_host_lib

pytestmark = pytest.mark.gpu

SECRET = 0x1F2E3D4C5B6A79887766554433221100AABBCCDDEEFF


@pytest.fixture(scope="module")
def H():
    L = load_host_lib()
    L.hd_plonk_verify.argtypes = [ctypes.c_int, ctypes.c_int, ctypes.c_char_p, ctypes.c_size_t, ctypes.c_char_p,
                                  ctypes.c_size_t, ctypes.c_char_p, ctypes.c_size_t, ctypes.c_uint32, ctypes.c_char_p,
                                  ctypes.c_char_p, ctypes.c_size_t, ctypes.POINTER(ctypes.c_uint32)]
    return L


DK = None


def dk_bytes():
    global DK
    if DK is None:
        DK = g1(O.G1_GEN) + O.g2_to_bytes(O.G2_GEN) + O.g2_to_bytes(O.g2_mul(O.G2_GEN, SECRET))
    return DK


def mk_transcript(kind, proof=None):
    cls = T.EvmTranscript if kind == 0 else T.PoseidonTranscript
    return cls() if proof is None else cls(proof)


def run(H, mos, kind, pr, instances_list, proofs):
    accs = ctypes.create_string_buffer(128 * 64 * max(1, len(proofs)))
    n = ctypes.c_uint32(0)
    pb = S.pack_protocol(pr)
    ib = b"".join(S.pack_instances(i) for i in instances_list)
    prb = b"".join(len(p).to_bytes(4, "little") + p for p in proofs)
    rc = H.hd_plonk_verify(mos, kind, pb, len(pb), ib, len(ib), prb, len(prb), len(proofs), dk_bytes(), accs, len(accs),
                           ctypes.byref(n))
    return rc, accs.raw[:128 * n.value]


def oracle_accs(mos_name, kind, pr, instances, proof):
    t = mk_transcript(kind, proof)
    pf = P.plonk_proof_read(pr, instances, t, mos_name)
    assert t.pos == len(t.stream)
    return P.succinct_verify(O.G1_GEN, pr, instances, pf, mos_name)


MOS = {0: "gwc19", 1: "bdfg21"}


@pytest.mark.parametrize("mos", [0, 1])
@pytest.mark.parametrize("kind", [0, 1])
@pytest.mark.parametrize("lin", [None, "WithoutConstant", "MinusVanishingTimesQuotient"])
def test_forged_proofs_verify_and_match_the_oracle(H, mos, kind, lin):
    rng = random.Random(1000 + 100 * mos + 10 * kind + (0 if lin is None else len(lin)))
    pr, dl = S.standard_plonk_protocol(rng, linearization=lin, num_instance=(2, 3))
    inst = [[rng.randrange(O.R) for _ in range(n)] for n in pr["num_instance"]]
    proof = P.forge_proof(pr, inst, SECRET, lambda: mk_transcript(kind), MOS[mos], rng, dl)
    exp = oracle_accs(MOS[mos], kind, pr, inst, proof)
    assert exp[0][0] == O.g1_mul(exp[0][1], SECRET)  # the forged proof really is valid under the toy SRS
    rc, accs = run(H, mos, kind, pr, [inst], [proof])
    assert rc == 1
    assert accs == b"".join(g1(a) + g1(b) for a, b in exp)
    # any single-bit change of the proof: rejected by the pairing, or already by the transcript
    for pos in (0, len(proof) // 2, len(proof) - 1):
        bad = bytearray(proof)
        bad[pos] ^= 1
        rc, _ = run(H, mos, kind, pr, [inst], [bytes(bad)])
        assert rc in (0, -10)
    # a changed instance changes every challenge: reject
    inst2 = [list(x) for x in inst]
    inst2[0][0] = (inst2[0][0] + 1) % O.R
    rc, _ = run(H, mos, kind, pr, [inst2], [proof])
    assert rc == 0


@pytest.mark.parametrize("mos", [0, 1])
@pytest.mark.parametrize("lin", ["WithoutConstant", "MinusVanishingTimesQuotient"])
def test_linearized_numerator_with_a_commitment_term(H, mos, lin):
    """A fixed column without an evaluation: its commitment flows through the
    numerator as an Msm (proof.rs:218-252), the linearization commitment carries it."""
    rng = random.Random(300 + mos + len(lin))
    pr, dl = S.standard_plonk_protocol(rng, linearization=lin, num_instance=(2, 2), linearize_fixed=True)
    inst = [[rng.randrange(O.R) for _ in range(n)] for n in pr["num_instance"]]
    proof = P.forge_proof(pr, inst, SECRET, lambda: mk_transcript(0), MOS[mos], rng, dl)
    exp = oracle_accs(MOS[mos], 0, pr, inst, proof)
    assert exp[0][0] == O.g1_mul(exp[0][1], SECRET)
    rc, accs = run(H, mos, 0, pr, [inst], [proof])
    assert rc == 1 and accs == b"".join(g1(a) + g1(b) for a, b in exp)


def test_old_accumulators_from_instance_limbs_and_committed_instances(H):
    rng = random.Random(7)
    a = rng.randrange(1, O.R)
    old = (O.g1_mul(O.G1_GEN, SECRET * a % O.R), O.g1_mul(O.G1_GEN, a))  # a valid accumulator to carry over
    limbs = K.accumulator_to_limbs(old)
    for committed in (False, True):
        pr, dl = S.standard_plonk_protocol(rng, num_instance=(17,), accumulator_rows=[list(range(16))],
                                           committed_instances=committed)
        inst = [limbs + [rng.randrange(O.R)]]
        proof = P.forge_proof(pr, inst, SECRET, lambda: mk_transcript(1), "bdfg21", rng, dl)
        exp = oracle_accs("bdfg21", 1, pr, inst, proof)
        assert len(exp) == 2 and exp[1] == old
        rc, accs = run(H, 1, 1, pr, [inst], [proof])
        assert rc == 1
        assert accs == b"".join(g1(x) + g1(y) for x, y in exp)
        # an INVALID old accumulator in the instances must fail the final decide_all
        bad_old = (old[0], O.g1_mul(O.G1_GEN, a + 1))
        inst_bad = [K.accumulator_to_limbs(bad_old) + inst[0][16:]]
        proof_bad = P.forge_proof(pr, inst_bad, SECRET, lambda: mk_transcript(1), "bdfg21", rng, dl)
        rc, _ = run(H, 1, 1, pr, [inst_bad], [proof_bad])
        assert rc == 0


def test_error_paths(H):
    rng = random.Random(9)
    pr, dl = S.standard_plonk_protocol(rng)
    inst = [[rng.randrange(O.R) for _ in range(n)] for n in pr["num_instance"]]
    proof = P.forge_proof(pr, inst, SECRET, lambda: mk_transcript(0), "gwc19", rng, dl)
    assert run(H, 0, 0, pr, [inst], [proof])[0] == 1
    # Error::InvalidInstances (proof.rs:67-74)
    assert run(H, 0, 0, pr, [[inst[0] + [1]]], [proof])[0] == -11
    # Error::Transcript: truncated proof
    assert run(H, 0, 0, pr, [inst], [proof[:-40]])[0] == -10
    # Error::InvalidProtocol("Missing challenge") (proof.rs:236-241)
    import copy
    pr2 = copy.deepcopy(pr)
    pr2["quotient"]["numerator"] = ("sum", pr["quotient"]["numerator"], ("challenge", 9))
    assert run(H, 0, 0, pr2, [inst], [proof])[0] == -12
    # Error::InvalidProtocol("Missing query")
    pr3 = copy.deepcopy(pr)
    pr3["quotient"]["numerator"] = ("sum", pr["quotient"]["numerator"], ("poly", 9, 5))
    assert run(H, 0, 0, pr3, [inst], [proof])[0] == -12
    # product of two commitments: Error::InvalidProtocol("Invalid linearization") (proof.rs:246-252)
    pr4 = copy.deepcopy(pr)
    pr4["evaluations"] = [q for q in pr["evaluations"] if q not in ((0, 0), (1, 0))]
    pr4["queries"] = [q for q in pr["queries"] if q not in ((0, 0), (1, 0))]
    pr4["quotient"]["numerator"] = ("sum", pr["quotient"]["numerator"], ("prod", ("poly", 0, 0), ("poly", 1, 0)))
    proof4 = P.forge_proof(pr, inst, SECRET, lambda: mk_transcript(0), "gwc19", rng, dl)
    assert run(H, 0, 0, pr4, [inst], [proof4])[0] in (-12, -10)


def test_batch_of_proofs_one_launch(H):
    rng = random.Random(11)
    pr, dl = S.standard_plonk_protocol(rng)
    insts, proofs, exp = [], [], b""
    for _ in range(12):
        inst = [[rng.randrange(O.R) for _ in range(n)] for n in pr["num_instance"]]
        proof = P.forge_proof(pr, inst, SECRET, lambda: mk_transcript(0), "gwc19", rng, dl)
        insts.append(inst)
        proofs.append(proof)
        exp += b"".join(g1(a) + g1(b) for a, b in oracle_accs("gwc19", 0, pr, inst, proof))
    rc, accs = run(H, 0, 0, pr, insts, proofs)
    assert rc == 1 and accs == exp
    bad = bytearray(proofs[5])
    bad[100] ^= 4
    proofs[5] = bytes(bad)
    assert run(H, 0, 0, pr, insts, proofs)[0] in (0, -10)


def test_golden_fixture_cpp(H):
    """tests/golden/plonk_forged.json (made by tests/golden/gen_golden_plonk.py): protocol
    bytes + instances + proof bytes in, accumulator bytes out, accept."""
    import json
    import os

    with open(os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden", "plonk_forged.json")) as f:
        g = json.load(f)
    for case in g["cases"]:
        mos = 0 if case["mos"] == "gwc19" else 1
        accs = ctypes.create_string_buffer(128 * 8)
        n = ctypes.c_uint32(0)
        pb, ib, proof = bytes.fromhex(case["protocol"]), bytes.fromhex(case["instances"]), bytes.fromhex(case["proof"])
        prb = len(proof).to_bytes(4, "little") + proof
        rc = H.hd_plonk_verify(mos, case["transcript"], pb, len(pb), ib, len(ib), prb, len(prb), 1, dk_bytes(), accs,
                               len(accs), ctypes.byref(n))
        assert rc == 1, case["name"]
        assert accs.raw[:128 * n.value].hex() == "".join(case["accumulators"]), case["name"]


def load_bench_blob(kind="evm"):
    import os
    import struct

    path = os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden", "bench_plonk_gwc19_%s_64.bin" % kind)
    b = open(path, "rb").read()
    assert b[:4] == b"SVB1"
    n, = struct.unpack_from("<I", b, 4)
    off = 8
    parts = []
    for _ in range(3):
        ln, = struct.unpack_from("<I", b, off)
        parts.append(b[off + 4:off + 4 + ln])
        off += 4 + ln
    dk, exp = b[off:off + 320], b[off + 320:off + 448]
    return n, parts[0], parts[1], parts[2], dk, exp


@pytest.mark.parametrize("threads", [1, 8])
def test_end_to_end_aggregation_of_64_proofs(H, threads):
    """BASELINE.json config C3 as real bytes (tests/golden/bench_plonk_gwc19_evm_64.bin):
    64 x PlonkSuccinctVerifier -> KzgAs::create_proof -> decide, host front half
    on `threads` threads, every EC operation on the device."""
    H.hd_aggregate_end_to_end.argtypes = [ctypes.c_int, ctypes.c_int, ctypes.c_char_p, ctypes.c_size_t, ctypes.c_char_p,
                                          ctypes.c_size_t, ctypes.c_char_p, ctypes.c_size_t, ctypes.c_uint32,
                                          ctypes.c_char_p, ctypes.c_uint, ctypes.POINTER(ctypes.c_double), ctypes.c_char_p]
    n, pb, ib, prb, dk, exp = load_bench_blob()
    tm = (ctypes.c_double * 6)()
    acc = ctypes.create_string_buffer(128)
    rc = H.hd_aggregate_end_to_end(0, 0, pb, len(pb), ib, len(ib), prb, len(prb), n, dk, threads, tm, acc)
    assert rc == 1
    assert acc.raw == exp  # the oracle's aggregated accumulator (gen_bench_proofs.py)
    # the same workload with Poseidon transcripts inside the proofs (the reference's recursion setting)
    n2, pb2, ib2, prb2, dk2, exp2 = load_bench_blob("poseidon")
    rc = H.hd_aggregate_end_to_end(0, 1, pb2, len(pb2), ib2, len(ib2), prb2, len(prb2), n2, dk2, threads, tm, acc)
    assert rc == 1 and acc.raw == exp2
    # ... and with the Poseidon hashing of all 64 transcripts in ONE device launch (csrc/poseidon.hip)
    acc2 = ctypes.create_string_buffer(128)
    rc = H.hd_aggregate_end_to_end(0, 2, pb2, len(pb2), ib2, len(ib2), prb2, len(prb2), n2, dk2, threads, tm, acc2)
    assert rc == 1 and acc2.raw == exp2
    bad2 = bytearray(prb2)
    bad2[4 + 40] ^= 1  # inside the first proof
    rc = H.hd_aggregate_end_to_end(0, 2, pb2, len(pb2), ib2, len(ib2), bytes(bad2), len(bad2), n2, dk2, threads, tm, acc2)
    assert rc in (0, -10)
    # replicate the batch 4x (256 proofs): still one launch per stage, still accepted
    rc = H.hd_aggregate_end_to_end(0, 0, pb, len(pb), ib * 4, 4 * len(ib), prb * 4, 4 * len(prb), 4 * n, dk, threads, tm, acc)
    assert rc == 1
    # a corrupted proof anywhere in the batch poisons the aggregate
    bad = bytearray(prb)
    first_len = int.from_bytes(prb[:4], "little")
    bad[4 + first_len + 4 + 700] ^= 2  # inside the payload of the second proof
    rc = H.hd_aggregate_end_to_end(0, 0, pb, len(pb), ib, len(ib), bytes(bad), len(bad), n, dk, threads, tm, acc)
    assert rc in (0, -10)


def test_proof_sharded_aggregation_wiring(H):
    """distributed.gpu_sharded_aggregation at world size 1 (the driver runs N > 1): succinct
    verify of the fixture's 64 proofs -> KzgAs -> decide through the host mirror, equal to the
    fixture's aggregated accumulator."""
    import struct

    from snark_verifier_amd import distributed as D

    from snark_verifier_amd import host_api as HA

    n, pb, ib, prb, dk, exp = load_bench_blob()
    insts, proofs, off = [], [], 0
    for _ in range(n):  # split the packed streams per proof
        cols, = struct.unpack_from("<I", ib, off)
        o2 = off + 4
        for _ in range(cols):
            m, = struct.unpack_from("<I", ib, o2)
            o2 += 4 + 32 * m
        insts.append(ib[off:o2])
        off = o2
    off = 0
    for _ in range(n):
        ln, = struct.unpack_from("<I", prb, off)
        proofs.append(prb[off + 4:off + 4 + ln])
        off += 4 + ln
    hp, hdk = HA.Protocol(pb), HA.DecidingKey(dk)
    acc, ok = D.gpu_sharded_aggregation(hp, hdk, insts, proofs)
    assert ok and acc == exp
    # a shard that fails to verify is a reject on every rank, not an exception that would leave the peers in a collective
    bad = list(proofs)
    bad[3] = bad[3][:100]
    assert D.gpu_sharded_aggregation(hp, hdk, insts, bad) == (None, False)
    hp.close()
    hdk.close()


@pytest.mark.parametrize("seed", list(range(24)))
def test_random_protocol_shapes_cpp_vs_oracle(H, seed):
    """Fuzzing the mirror against the oracle: random polynomial counts, phases, rotations,
    expression trees over every node kind (nested DistributePowers, Scaled, Negated, ...),
    quotient chunking, both multi-open schemes, both transcripts, all linearization modes."""
    rng = random.Random(5000 + seed)
    lin = rng.choice([None, None, "WithoutConstant", "MinusVanishingTimesQuotient"])
    mos = rng.randrange(2)
    kind = rng.randrange(2)
    pr, dl = S.random_protocol(rng, lin)
    inst = [[rng.randrange(O.R) for _ in range(n)] for n in pr["num_instance"]]
    proof = P.forge_proof(pr, inst, SECRET, lambda: mk_transcript(kind), MOS[mos], rng, dl)
    exp = oracle_accs(MOS[mos], kind, pr, inst, proof)
    assert exp[0][0] == O.g1_mul(exp[0][1], SECRET)
    rc, accs = run(H, mos, kind, pr, [inst], [proof])
    assert rc == 1, (seed, lin, mos, kind)
    assert accs == b"".join(g1(a) + g1(b) for a, b in exp)


def test_poseidon_auto_transcript_picks_by_batch_size():
    """SNARKV_HOST_TRANSCRIPT_POSEIDON_AUTO (include/snarkv_host.h): host-hashed below SNARKV_HOST_POSEIDON_DEVICE_MIN
    proofs, device-hashed from there on -- the same accumulator and verdict as either explicit kind on both sides of
    the threshold (64 proofs; the same 64 replicated to 512 and to 1 024: the threshold is 512 on the scalar sponge and
    1 536 where the host sponge runs on AVX-512 IFMA; ONE job on a host with 32+ pool threads stays on the host, pipelined,
    from 256 proofs on)."""
    from snark_verifier_amd import host_api as HA

    fx = HA.read_fixture(os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "tests", "golden",
                                      "bench_plonk_gwc19_poseidon_64.bin"))
    hp, hdk = HA.Protocol(fx["protocol"]), HA.DecidingKey(fx["dk"])
    for rep in (1, 8, 16):
        got = {}
        for kind in (HA.TRANSCRIPT_POSEIDON, HA.TRANSCRIPT_POSEIDON_DEVICE, HA.TRANSCRIPT_POSEIDON_AUTO):
            ok, acc, tm = HA.aggregate(hp, hdk, fx["instances"] * rep, fx["proofs"] * rep, fx["n"] * rep, HA.MOS_GWC19, kind, 8,
                                       timings=True)
            assert ok
            got[kind] = acc
        assert len(set(got.values())) == 1
        if rep == 1:
            assert got[HA.TRANSCRIPT_POSEIDON_AUTO] == fx["expected_acc"]
    hp.close()
    hdk.close()


def test_device_decompressed_points_keep_the_host_verdicts():
    """The device-hashed Poseidon path decompresses the whole batch's points in one launch (`bn254_g1_decompress`) and
    hands them to the parsing pass as hints that are checked against the bytes.  Tampered batches must end exactly as
    on the host-hashed path: a corrupted x (another point, or no point at all), a flipped parity bit, a proof of another
    length (which keeps the host path) -- same verdict, same error text, same accumulator."""
    import struct

    from snark_verifier_amd import host_api as HA

    fx = HA.read_fixture(os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "tests", "golden",
                                      "bench_plonk_gwc19_poseidon_64.bin"))
    hp, hdk = HA.Protocol(fx["protocol"]), HA.DecidingKey(fx["dk"])
    blob = fx["proofs"]
    starts, off = [], 0
    for _ in range(fx["n"]):
        ln, = struct.unpack_from("<I", blob, off)
        starts.append((off + 4, ln))
        off += 4 + ln

    def outcome(proofs, kind):
        try:
            ok, acc = HA.aggregate(hp, hdk, fx["instances"], proofs, fx["n"], HA.MOS_GWC19, kind, 8)
            return ("rc", ok, acc if ok else None)
        except HA.HostError as e:
            return ("err", e.code if hasattr(e, "code") else None, str(e))

    def both(proofs):
        a, b = outcome(proofs, HA.TRANSCRIPT_POSEIDON), outcome(proofs, HA.TRANSCRIPT_POSEIDON_DEVICE)
        assert a == b, (a, b)
        return a

    assert both(blob) == ("rc", True, fx["expected_acc"])
    s5 = starts[5][0]
    for mutate in (lambda m: m.__setitem__(s5, m[s5] ^ 1),                 # x of the first point of proof 5
                   lambda m: m.__setitem__(s5 + 31, m[s5 + 31] ^ 0x40),    # its parity bit: the other root
                   lambda m: m.__setitem__(s5 + 31, m[s5 + 31] | 0x80),    # an identity flag on a finite x
                   lambda m: m.__setitem__(starts[63][0] + 32 + 7, m[starts[63][0] + 32 + 7] ^ 0x55)):  # second point, last proof
        m = bytearray(blob)
        mutate(m)
        r = both(bytes(m))
        assert r != ("rc", True, fx["expected_acc"])
    # a proof that is one scalar longer: another length -> no device hints for it, the same ending on both paths
    s9, l9 = starts[9]
    longer = blob[:s9 - 4] + struct.pack("<I", l9 + 32) + blob[s9:s9 + l9] + bytes(32) + blob[s9 + l9:]
    both(longer)
    hp.close()
    hdk.close()


def test_concurrent_aggregations_from_host_threads():
    """Several host threads in `snarkv_host_aggregate` at once (ctypes releases the GIL): the device lock is held while
    the terms are packed into the default context's pinned buffers (loader.hpp), the host pool runs one job at a time,
    the Poseidon path takes the device for its hashing and decompression launches -- every call must still return the
    fixture's accumulator, whatever the interleaving."""
    import threading

    from snark_verifier_amd import host_api as HA

    root = os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "tests", "golden")
    jobs = []
    for name, kind in (("bench_plonk_gwc19_evm_64.bin", HA.TRANSCRIPT_EVM),
                       ("bench_plonk_gwc19_poseidon_64.bin", HA.TRANSCRIPT_POSEIDON_DEVICE),
                       ("bench_plonk_gwc19_poseidon_64.bin", HA.TRANSCRIPT_POSEIDON)):
        fx = HA.read_fixture(os.path.join(root, name))
        jobs.append((fx, HA.Protocol(fx["protocol"]), HA.DecidingKey(fx["dk"]), kind))
    bad = []

    def worker(k):
        fx, hp, hdk, kind = jobs[k % len(jobs)]
        for _ in range(4):
            try:
                ok, acc = HA.aggregate(hp, hdk, fx["instances"], fx["proofs"], fx["n"], HA.MOS_GWC19, kind, 8)
                if not ok or acc != fx["expected_acc"]:
                    bad.append((k, "wrong result"))
            except Exception as e:  # noqa: BLE001
                bad.append((k, repr(e)))

    threads = [threading.Thread(target=worker, args=(k,)) for k in range(6)]
    for t in threads:
        t.start()
    for t in threads:
        t.join(timeout=120)
    assert not any(t.is_alive() for t in threads), "a worker is stuck (lock order?)"
    assert not bad, bad
    for _, hp, hdk, _ in jobs:
        hp.close()
        hdk.close()


def test_aggregate_many_equals_one_call_per_job():
    """`snarkv_host_aggregate_many`: several jobs sharing their three device launches give, per job, the accumulator and
    the verdict of `snarkv_host_aggregate` on that job's proofs -- for ragged job sizes, both Keccak and Poseidon
    transcripts, and with one job made to fail its pairing check (a proof swapped for another job's valid one changes
    nothing; a tampered evaluation does)."""
    import struct

    from snark_verifier_amd import host_api as HA

    root = os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "tests", "golden")
    for name, kind in (("bench_plonk_gwc19_evm_64.bin", HA.TRANSCRIPT_EVM),
                       ("bench_plonk_gwc19_poseidon_64.bin", HA.TRANSCRIPT_POSEIDON_DEVICE),
                       ("bench_plonk_gwc19_poseidon_64.bin", HA.TRANSCRIPT_POSEIDON)):
        fx = HA.read_fixture(os.path.join(root, name))
        hp, hdk = HA.Protocol(fx["protocol"]), HA.DecidingKey(fx["dk"])
        # split the blobs per proof
        pblob, iblob = fx["proofs"], fx["instances"]
        proofs, off = [], 0
        for _ in range(fx["n"]):
            ln, = struct.unpack_from("<I", pblob, off)
            proofs.append(pblob[off:off + 4 + ln])
            off += 4 + ln
        insts, off = [], 0
        for _ in range(fx["n"]):
            start = off
            cols, = struct.unpack_from("<I", iblob, off)
            off += 4
            for _c in range(cols):
                m, = struct.unpack_from("<I", iblob, off)
                off += 4 + 32 * m
            insts.append(iblob[start:off])
        assert off == len(iblob)
        sizes = [5, 1, 30, 28]
        assert sum(sizes) == fx["n"]
        # tamper one scalar of a proof of job 2 (the last 32 bytes of a Gwc19 proof are an opening point or scalar:
        # flip a byte in the middle of the proof instead, inside the evaluations)
        bad = bytearray(proofs[10])
        bad[4 + len(bad) // 2] ^= 1
        variants = {"clean": list(proofs), "tampered": proofs[:10] + [bytes(bad)] + proofs[11:]}
        for label, plist in variants.items():
            want, first = [], 0
            for k in sizes:
                try:
                    ok, acc = HA.aggregate(hp, hdk, b"".join(insts[first:first + k]), b"".join(plist[first:first + k]), k,
                                           HA.MOS_GWC19, kind, 8)
                    want.append((ok, acc))
                except HA.HostError as e:
                    want.append(("err", str(e)))
                first += k
            if any(w[0] == "err" for w in want):  # a malformed proof fails the whole call, as it fails its own job
                with pytest.raises(HA.HostError):
                    HA.aggregate_many(hp, hdk, b"".join(insts), b"".join(plist), sizes, HA.MOS_GWC19, kind, 8)
                continue
            allok, accs, oks = HA.aggregate_many(hp, hdk, b"".join(insts), b"".join(plist), sizes, HA.MOS_GWC19, kind, 8)
            assert oks == [w[0] for w in want], (name, label)
            assert accs == [w[1] for w in want], (name, label)
            assert allok == all(oks)
            if label == "clean":
                assert allok
        # the sizes must add up
        with pytest.raises(HA.HostError):
            HA.aggregate_many(hp, hdk, b"".join(insts), b"".join(proofs), [5, 1, 30, 0, 28], HA.MOS_GWC19, kind, 8)
        hp.close()
        hdk.close()


def test_host_hashed_poseidon_with_and_without_device_hints(monkeypatch):
    """Round 5: batches of >= SNARKV_HOST_HINT_MIN (default: 32, and more than two proofs per host thread -- 8 threads here)
    host-hashed Poseidon proofs get their compressed points decompressed by
    ONE device launch and offered to the transcripts as hints (checked against the bytes before use).  Same accumulator with
    the hints, without them (threshold raised), and for a batch with one proof of another length (no hint for that one,
    Error::Transcript from the host path as before)."""
    from snark_verifier_amd import host_api as HA

    fx = HA.read_fixture(os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "tests", "golden",
                                      "bench_plonk_gwc19_poseidon_64.bin"))
    hp, hdk = HA.Protocol(fx["protocol"]), HA.DecidingKey(fx["dk"])
    monkeypatch.setenv("SNARKV_HOST_HINT_MIN", "2")  # (the default on a CPU with AVX-512 IFMA is "never": transcript.hpp g1_decompress_x8)
    ok, with_hints = HA.aggregate(hp, hdk, fx["instances"], fx["proofs"], fx["n"], HA.MOS_GWC19, HA.TRANSCRIPT_POSEIDON, 8)
    assert ok and with_hints == fx["expected_acc"]
    # ... without device hints and without the transcript's own grouped decoding: the scalar square roots
    monkeypatch.setenv("SNARKV_HOST_HINT_MIN", "100000")
    monkeypatch.setenv("SNARKV_HOST_NO_POINT_PREFETCH", "1")
    ok, scalar = HA.aggregate(hp, hdk, fx["instances"], fx["proofs"], fx["n"], HA.MOS_GWC19, HA.TRANSCRIPT_POSEIDON, 8)
    assert ok and scalar == fx["expected_acc"]
    monkeypatch.delenv("SNARKV_HOST_NO_POINT_PREFETCH")
    ok, without = HA.aggregate(hp, hdk, fx["instances"], fx["proofs"], fx["n"], HA.MOS_GWC19, HA.TRANSCRIPT_POSEIDON, 8)
    assert ok and without == fx["expected_acc"]
    monkeypatch.delenv("SNARKV_HOST_HINT_MIN")
    # a corrupted point in proof 40 (its first commitment's x made a non-residue candidate by flipping a low bit until the
    # oracle's decompression fails): rejected with the hints in play exactly as without
    import transcript as T

    prb = bytearray(fx["proofs"])
    off = 0
    for _ in range(40):
        off += 4 + int.from_bytes(prb[off:off + 4], "little")
    for bit in range(8):
        cand = bytearray(prb)
        cand[off + 4] ^= 1 << bit
        try:
            T.g1_decompress(bytes(cand[off + 4:off + 36]))
        except T.TranscriptError:
            break
    else:
        pytest.skip("no invalid encoding found by flipping one byte")
    for hint_min in ("2", "100000"):
        monkeypatch.setenv("SNARKV_HOST_HINT_MIN", hint_min)
        with pytest.raises(HA.HostError) as e:
            HA.aggregate(hp, hdk, fx["instances"], bytes(cand), fx["n"], HA.MOS_GWC19, HA.TRANSCRIPT_POSEIDON, 8)
        assert e.value.code == HA.ERR_TRANSCRIPT
    hp.close()
    hdk.close()
