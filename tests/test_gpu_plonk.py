"""GPU: the PLONK succinct verifier front half of the host mirror
(snark-verifier_amd/host/plonk.hpp; reference verifier/plonk.rs:58-147,
verifier/plonk/proof.rs:52-349, verifier/plonk/protocol.rs) end to end:
proof bytes -> transcript -> expression evaluation -> MSMs on the MI355X ->
pairing decide, against oracle/plonk.py on proofs FORGED under a toy SRS
(no halo2 prover exists here; see oracle/plonk.py)."""
import ctypes
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
