"""CPU: the pallas oracle (oracle/pallas.py), halo2's Blake2b transcript (oracle/transcript.py) and the
curve-generic IPA oracle run on pallas -- the reference's `test_ipa` / `test_ipa_as`
(pcs/ipa.rs:434-466, pcs/ipa/accumulation.rs:240-290) on their own curve and transcript, seeded."""
import os
import random
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "oracle"))

import bn254 as BN  # noqa: E402
import ipa as I  # noqa: E402
import pallas as PA  # noqa: E402
import transcript as T  # noqa: E402


@pytest.fixture()
def on_pallas():
    I.use_curve(PA)
    yield
    I.use_curve(BN)


def test_curve_constants_and_group_law():
    assert PA.P % 4 == 1 and (PA.P - 1) % (1 << 32) == 0  # the 2-adicity halo2 picks the curve for
    assert PA.g1_is_on_curve(PA.G1_GEN) and PA.G1_GEN == (PA.P - 1, 2)
    assert PA.g1_mul(PA.G1_GEN, PA.R) is None                       # prime order r
    assert PA.g1_mul(PA.G1_GEN, PA.R - 1) == PA.g1_neg(PA.G1_GEN)
    rnd = random.Random(2)
    pts = PA.sample_points(3, 20)
    assert all(PA.g1_is_on_curve(p) for p in pts) and len(set(pts)) == 20
    a, b = rnd.randrange(PA.R), rnd.randrange(PA.R)
    p = pts[0]
    assert PA.g1_add(PA.g1_mul(p, a), PA.g1_mul(p, b)) == PA.g1_mul(p, (a + b) % PA.R)
    assert PA.g1_add(p, PA.g1_neg(p)) is None and PA.g1_add(p, p) == PA.g1_double(p) == PA.g1_mul(p, 2)
    assert PA.g1_from_bytes(PA.g1_to_bytes(p)) == p and PA.g1_from_bytes(bytes(64)) is None
    sc = [rnd.randrange(PA.R) for _ in pts]
    sc[3], sc[4] = 0, PA.R - 1
    assert PA.g1_msm_naive(sc, pts) == PA.g1_msm_pippenger(sc, pts) == PA.g1_msm_pippenger(sc, pts, c=5)
    y = PA.fq_sqrt(9)
    assert y in (3, PA.P - 3) and PA.fq_sqrt(PA.P - 1) is not None  # -1 is a square: p = 1 mod 4


def test_blake2b_transcript_round_trip_and_framing():
    import hashlib

    rnd = random.Random(5)
    pts = PA.sample_points(9, 3)
    w = T.Blake2bTranscript(PA)
    w.write_ec_point(pts[0])
    c1 = w.squeeze_challenge()
    s = rnd.randrange(PA.R)
    w.write_scalar(s)
    w.write_ec_point((pts[1][0], PA.P - pts[1][1]))  # the other root: the sign bit must carry it
    c2 = w.squeeze_challenge()
    proof = w.finalize()
    assert len(proof) == 96
    r = T.Blake2bTranscript(PA, proof)
    assert r.read_ec_point() == pts[0] and r.squeeze_challenge() == c1
    assert r.read_scalar() == s and r.read_ec_point() == (pts[1][0], PA.P - pts[1][1]) and r.squeeze_challenge() == c2
    with pytest.raises(T.TranscriptError):
        r.read_scalar()
    # framing, written out: personalised BLAKE2b-512 over 0x01 || x || y, then 0x00, digest of a copy, LE mod r
    h = hashlib.blake2b(digest_size=64, person=b"Halo2-Transcript")
    h.update(b"\x01" + PA.fe_to_bytes(pts[0][0]) + PA.fe_to_bytes(pts[0][1]) + b"\x00")
    assert c1 == int.from_bytes(h.digest(), "little") % PA.R
    with pytest.raises(T.TranscriptError):
        T.Blake2bTranscript(PA, (PA.R).to_bytes(32, "little")).read_scalar()  # non-canonical scalar
    with pytest.raises(T.TranscriptError):
        T.Blake2bTranscript(PA, bytes(32)).read_ec_point()                    # the identity never travels


@pytest.mark.parametrize("zk", [False, True])
def test_ipa_on_pallas_like_the_reference_tests(on_pallas, zk):
    rnd = random.Random("pallas-%d" % zk)
    rng = lambda: rnd.randrange(PA.R)  # noqa: E731
    k = 4
    pts = PA.sample_points(77 + zk, (1 << k) + 2)
    pk = I.IpaProvingKey(k, pts[:1 << k], pts[1 << k], pts[(1 << k) + 1] if zk else None)
    accs = []
    for _ in range(3):
        p = [rng() for _ in range(1 << k)]
        omega, z = (rng() if zk else None), rng()
        c = pk.commit(p, omega)
        t = T.Blake2bTranscript(PA)
        acc = I.ipa_create_proof(pk, p, z, omega, t, rng)
        proof = t.finalize()
        got = I.ipa_succinct_verify(pk.h, pk.s, [(1, c)], z, I.poly_eval(p, z),
                                    I.ipa_read_proof(zk, k, T.Blake2bTranscript(PA, proof)))
        assert got == acc and I.ipa_decide(pk.g, acc)
        assert not I.ipa_decide(pk.g, (acc[0], PA.g1_add(acc[1], pk.h)))
        with pytest.raises(I.IpaError):
            I.ipa_succinct_verify(pk.h, pk.s, [(1, c)], z, (I.poly_eval(p, z) + 1) % PA.R,
                                  I.ipa_read_proof(zk, k, T.Blake2bTranscript(PA, proof)))
        accs.append(acc)
    t = T.Blake2bTranscript(PA)
    acc = I.ipa_as_create_proof(pk, accs, t, rng)
    got = I.ipa_as_verify(pk.h, pk.s, accs, I.ipa_as_read_proof(zk, k, accs, T.Blake2bTranscript(PA, t.finalize())))
    assert got == acc and I.ipa_decide(pk.g, acc)


def test_use_curve_switches_back():
    I.use_curve(PA)
    I.use_curve(BN)
    assert I.R == BN.R and I._FAST
