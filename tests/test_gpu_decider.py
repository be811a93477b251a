"""GPU parity: the pairing decider kernels vs the oracle.
`KzgAs::decide` / `decide_all` (reference snark-verifier/src/pcs/kzg/decider.rs:70-93)."""
import pytest

import bn254 as O

pytestmark = pytest.mark.gpu


@pytest.fixture(scope="module")
def dk(gpu_ctx, golden_decider):
    import snark_verifier_amd as sv

    g = golden_decider
    k = sv.DecidingKey(gpu_ctx, bytes.fromhex(g["g1"]), bytes.fromhex(g["g2"]), bytes.fromhex(g["s_g2"]),
                       flags=sv.SNARKV_FLAG_VALIDATE)
    yield k
    k.close()


@pytest.mark.parametrize("teams", ["1", "3"])
def test_golden_decide_and_gt_value(gpu_ctx, dk, golden_decider, teams, monkeypatch):
    """Both kernel forms (decider.hip: one-team throughput form, two-team latency
    form) must give the exact Gt element, identity pairs included."""
    monkeypatch.setenv("SNARKV_DECIDE_FORM", teams)
    for case in golden_decider["cases"]:
        acc = bytes.fromhex(case["acc"])
        assert gpu_ctx.decide(dk, acc) == case["accept"], case["name"]
        # stronger than the boolean: the exact Gt element (final exponent exact)
        assert gpu_ctx.pairing_value(dk, acc) == bytes.fromhex(case["gt"]), case["name"]


@pytest.mark.parametrize("teams", ["1", "3"])
def test_decide_all_batch(gpu_ctx, dk, golden_decider, teams, monkeypatch):
    monkeypatch.setenv("SNARKV_DECIDE_FORM", teams)
    cases = golden_decider["cases"]
    accs = b"".join(bytes.fromhex(c["acc"]) for c in cases)
    allok, oks = gpu_ctx.decide_batch(dk, accs)
    assert oks == [c["accept"] for c in cases]
    assert allok == all(c["accept"] for c in cases)
    valid = [bytes.fromhex(c["acc"]) for c in cases if c["accept"]]
    allok, oks = gpu_ctx.decide_batch(dk, b"".join(valid * 40))
    assert allok and all(oks) and len(oks) == 40 * len(valid)
    assert gpu_ctx.decide_batch(dk, b"") == (True, [])  # decide_all(vec![]) is Ok(())


def test_mock_accumulator_like_reference_fixture(gpu_ctx, dk, golden_decider):
    """The reference's mock-valid accumulator idea
    (snark-verifier/src/system/halo2/test/kzg.rs:30-46): (lhs, rhs) = (s*G, G)."""
    s = int(golden_decider["secret"], 16)
    lhs, rhs = O.g1_mul(O.G1_GEN, s), O.G1_GEN
    acc = O.g1_to_bytes(lhs) + O.g1_to_bytes(rhs)
    assert gpu_ctx.decide(dk, acc)
    for bit in (0, 100, 255, 256, 300, 511):  # single-bit perturbations of the x/y words must reject
        bad = bytearray(acc)
        bad[bit // 8] ^= 1 << (bit % 8)
        # perturbed coordinates are (almost surely) off-curve: either way not accepted
        assert not gpu_ctx.decide(dk, bytes(bad))
    other = O.g1_to_bytes(O.g1_mul(O.G1_GEN, s + 1)) + O.g1_to_bytes(rhs)
    assert not gpu_ctx.decide(dk, other)


def test_validation_flags(gpu_ctx, golden_decider):
    import snark_verifier_amd as sv

    g = golden_decider
    bad_g2 = bytearray(bytes.fromhex(g["g2"]))
    bad_g2[3] ^= 4
    with pytest.raises(sv.SnarkvError) as e:
        sv.DecidingKey(gpu_ctx, bytes.fromhex(g["g1"]), bytes(bad_g2), bytes.fromhex(g["s_g2"]),
                       flags=sv.SNARKV_FLAG_VALIDATE)
    assert e.value.code == -3
