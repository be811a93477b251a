"""BASELINE.json configs[4] at size under `-m gpu`: 1 024 DISTINCT proofs through the whole native pipeline
(the call pattern of snark-verifier/examples/evm-verifier-with-accumulator.rs:357-385, x1024) and the pairing
decider over 1 024 distinct accumulators (`decide_all`, pcs/kzg/decider.rs:84-93), against what the oracle
computed for the committed fixtures tests/golden/bench_plonk_gwc19_{evm,poseidon}_1024.bin (gen_bench_proofs.py 1024)
and against the C oracle's pairing on the box's host cores.  Everything goes through the C APIs
(include/snarkv_host.h over include/snarkv_amd.h)."""
import os

import pytest

import bn254 as O
import coracle as C
from snark_verifier_amd import host_api as H

pytestmark = pytest.mark.gpu
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


# the C5 workload on BOTH transcripts: Keccak proofs (the outer EVM flow) and Poseidon proofs with a Poseidon accumulation
# transcript -- the reference example's own native route (evm-verifier-with-accumulator.rs:361,375).  For Poseidon every
# route of the host mirror must give the same bytes: hashed on the host, on the device (one fused pipeline), auto.
FIXTURES = {"evm": ("bench_plonk_gwc19_evm_1024.bin", H.MOS_GWC19, [H.TRANSCRIPT_EVM]),
            "poseidon": ("bench_plonk_gwc19_poseidon_1024.bin", H.MOS_GWC19,
                         [H.TRANSCRIPT_POSEIDON_DEVICE, H.TRANSCRIPT_POSEIDON, H.TRANSCRIPT_POSEIDON_AUTO])}


def _load(kind):
    f = H.read_fixture(os.path.join(ROOT, "tests", "golden", FIXTURES[kind][0]))
    f["mos"], f["tkinds"], f["kind"] = FIXTURES[kind][1], FIXTURES[kind][2], kind
    return f


@pytest.fixture(scope="module")
def fx():
    return _load("evm")


@pytest.fixture(scope="module", params=["evm", "poseidon"])
def fxk(request):
    return _load(request.param)


@pytest.fixture(scope="module")
def handles(fx):
    hp, hdk = H.Protocol(fx["protocol"]), H.DecidingKey(fx["dk"])
    yield hp, hdk
    hp.close()
    hdk.close()


@pytest.fixture(scope="module")
def handlesk(fxk):
    hp, hdk = H.Protocol(fxk["protocol"]), H.DecidingKey(fxk["dk"])
    yield hp, hdk
    hp.close()
    hdk.close()


def test_1024_distinct_proofs_per_proof_accumulators_equal_oracle(fxk, handlesk):
    hp, hdk = handlesk
    assert fxk["n"] == 1024 and len(fxk["accs"]) == 128 * 1024
    for tk in fxk["tkinds"]:
        accs = H.plonk_succinct_verify_batch(hp, hdk, fxk["instances"], fxk["proofs"], fxk["n"], fxk["mos"], tk, strict=True)
        assert accs == fxk["accs"], tk  # 1 024 x (lhs, rhs), byte for byte what oracle/plonk.py's succinct verifier computed
    assert len({accs[128 * i:128 * i + 128] for i in range(1024)}) == 1024


def test_1024_distinct_proofs_aggregate_and_decide(fxk, handlesk):
    hp, hdk = handlesk
    for tk in fxk["tkinds"]:
        ok, acc = H.aggregate(hp, hdk, fxk["instances"], fxk["proofs"], fxk["n"], fxk["mos"], tk)
        # KzgAs over 1 024 accumulators on a transcript of the proofs' family == oracle/kzg.py's, then the pairing accepts
        assert ok and acc == fxk["expected_acc"], tk
    if fxk["kind"] == "evm":  # the two halves separately: accumulate the fixture's accumulators (Keccak), decide the result
        acc2, r = H.kzg_as_accumulate(fxk["accs"])
        assert acc2 == fxk["expected_acc"]
    else:  # `As::create_proof` on a Poseidon transcript through its own C API
        acc2, as_proof, r = H.kzg_as_create_proof(fxk["accs"], H.TRANSCRIPT_POSEIDON)
        assert acc2 == fxk["expected_acc"] and as_proof == b""  # non-zk: the proof carries nothing (accumulation.rs:181-196)
    assert H.kzg_decide(hdk, acc2)
    # the CPU restatement of the decider agrees on that accumulator (C oracle pairing, decider.rs:70-82)
    assert C.kzg_decide(fxk["dk"][64:192], fxk["dk"][192:320], acc2)
    # PlonkVerifier::verify on all 1 024: succinct verify + ONE decide_all over 1 024 accumulators
    assert H.plonk_verify(hp, hdk, fxk["instances"], fxk["proofs"], fxk["n"], fxk["mos"], fxk["tkinds"][0])
    # 16 jobs of 64 proofs in one call: every job's accumulator = one call on its 64 proofs
    oka, accs, oks = H.aggregate_many(hp, hdk, fxk["instances"], fxk["proofs"], [64] * 16, fxk["mos"], fxk["tkinds"][0])
    assert oka and all(oks) and len(set(accs)) == 16


def test_1024_proofs_with_a_corrupted_one_reject(fxk, handlesk):
    hp, hdk = handlesk
    prb = bytearray(fxk["proofs"])
    # flip one bit inside the 700th proof's evaluation section: still parses, no longer verifies
    off = 0
    for _ in range(700):
        off += 4 + int.from_bytes(prb[off:off + 4], "little")
    ln = int.from_bytes(prb[off:off + 4], "little")
    prb[off + 4 + ln - 200] ^= 4
    for tk in fxk["tkinds"]:
        try:
            ok, _ = H.aggregate(hp, hdk, fxk["instances"], bytes(prb), fxk["n"], fxk["mos"], tk)
            assert not ok, tk
        except H.HostError as e:  # or the flipped byte made a non-canonical scalar: Error::Transcript
            assert e.code == H.ERR_TRANSCRIPT


def _proof_offsets(blob, n):
    offs, off = [], 0
    for _ in range(n):
        ln = int.from_bytes(blob[off:off + 4], "little")
        offs.append((off + 4, ln))
        off += 4 + ln
    return offs


def _outcome(call):
    try:
        ok, acc = call()
        return ("ok", ok, acc)
    except H.HostError as e:
        return ("err", e.code, str(e))


def test_pipelined_poseidon_job_is_the_unpipelined_job_bit_for_bit(monkeypatch):
    """The pipelined form of a large Poseidon job (host/aggregation.hpp `aggregate_pipelined`: reading, MSMs and the
    accumulation sponge overlapped chunk by chunk) against the same job with the pipeline switched off and against the
    oracle's accumulator: ragged last chunks, one chunk, one proof per chunk-thread; and a bad proof in the first, a
    middle and the last chunk ends the job with the unpipelined job's outcome (first error in proof order)."""
    f = _load("poseidon")
    hp, hdk = H.Protocol(f["protocol"]), H.DecidingKey(f["dk"])
    n, tk = f["n"], H.TRANSCRIPT_POSEIDON
    monkeypatch.setenv("SNARKV_HOST_PIPELINE_MIN", "0")  # never
    ok0, acc0, tm0 = H.aggregate(hp, hdk, f["instances"], f["proofs"], n, f["mos"], tk, timings=True)
    assert ok0 and acc0 == f["expected_acc"]
    for chunk, pmin in (("128", "256"), ("100", "2"), ("1000", "2"), ("4096", "2"), ("37", "2")):
        monkeypatch.setenv("SNARKV_HOST_PIPELINE_MIN", pmin)
        monkeypatch.setenv("SNARKV_HOST_PIPELINE_CHUNK", chunk)
        ok, acc, tm = H.aggregate(hp, hdk, f["instances"], f["proofs"], n, f["mos"], tk, timings=True)
        assert ok and acc == acc0, chunk
        # the phases of the pipelined job overlap: the helpers' busy times run under `accumulate`; total is wall time
        assert tm["total"] > 0 and tm["accumulate"] <= tm["total"] and tm["read_proofs"] > 0 and tm["msm_device"] > 0
    monkeypatch.setenv("SNARKV_HOST_PIPELINE_CHUNK", "128")
    # a sub-batch (n < the fixture's 1 024: its own accumulator) and a batch below the threshold
    offs = _proof_offsets(f["proofs"], n)
    ioffs, off = [], 0
    inst = f["instances"]
    for _ in range(n):  # per proof: u32 columns, per column u32 m, m x Fr
        start = off
        cols = int.from_bytes(inst[off:off + 4], "little")
        off += 4
        for _c in range(cols):
            m = int.from_bytes(inst[off:off + 4], "little")
            off += 4 + 32 * m
        ioffs.append((start, off))
    for m in (300, 257):
        sub_p, sub_i = f["proofs"][:offs[m][0] - 4], inst[:ioffs[m][0]]
        monkeypatch.setenv("SNARKV_HOST_PIPELINE_MIN", "0")
        a = _outcome(lambda: H.aggregate(hp, hdk, sub_i, sub_p, m, f["mos"], tk))
        monkeypatch.setenv("SNARKV_HOST_PIPELINE_MIN", "256")
        b = _outcome(lambda: H.aggregate(hp, hdk, sub_i, sub_p, m, f["mos"], tk))
        assert a == b and a[0] == "ok" and a[1] is True, m
    # failures: truncate-free corruptions at chosen proofs (a flipped evaluation bit -> reject or Error::Transcript; a
    # broken compressed point -> Error::Transcript), one per chunk position, and two at once (the FIRST one decides)
    for bad in ([5], [700], [1023], [900, 130]):
        prb = bytearray(f["proofs"])
        for j, i in enumerate(bad):
            o, ln = offs[i]
            if j % 2 == 0:
                prb[o + ln - 200] ^= 4
            else:
                prb[o + 31] ^= 0x3F  # the flag / top bits of the first compressed point
        outs = []
        for pmin in ("0", "256"):
            monkeypatch.setenv("SNARKV_HOST_PIPELINE_MIN", pmin)
            outs.append(_outcome(lambda: H.aggregate(hp, hdk, f["instances"], bytes(prb), n, f["mos"], tk)))
        assert outs[0] == outs[1], (bad, outs)
        assert outs[0][0] == "err" or outs[0][1] is False, bad
    hp.close()
    hdk.close()


def test_shplonk_shape_64_proofs_bdfg21_poseidon_every_route():
    """The SDK's default scheme: SHPLONK = KzgAs<Bn256, Bdfg21> on Poseidon transcripts (snark-verifier-sdk/src/lib.rs:41,
    src/halo2.rs:296-306): 64 proofs of the committed fixture.  Bdfg21 reads W' AFTER its last squeeze (bdfg21.rs:64-66), so
    the transcript has trailing absorbs that feed no challenge -- the case that made the device-hashed routes abort
    (ADVICE r4): every route now gives the oracle's bytes."""
    f = H.read_fixture(os.path.join(ROOT, "tests", "golden", "bench_plonk_bdfg21_poseidon_64.bin"))
    hp, hdk = H.Protocol(f["protocol"]), H.DecidingKey(f["dk"])
    for tk in (H.TRANSCRIPT_POSEIDON, H.TRANSCRIPT_POSEIDON_DEVICE, H.TRANSCRIPT_POSEIDON_AUTO):
        accs = H.plonk_succinct_verify_batch(hp, hdk, f["instances"], f["proofs"], f["n"], H.MOS_BDFG21, tk, strict=True)
        assert accs == f["accs"], tk
        ok, acc = H.aggregate(hp, hdk, f["instances"], f["proofs"], f["n"], H.MOS_BDFG21, tk)
        assert ok and acc == f["expected_acc"], tk
    # proofs of DIFFERENT lengths cannot take the fused pipeline: drop the last byte of one proof -> Error::Transcript from
    # the three-pass route, not a crash
    prb = bytearray(f["proofs"])
    ln0 = int.from_bytes(prb[:4], "little")
    cut = prb[:4 + ln0 - 1]
    cut[:4] = (ln0 - 1).to_bytes(4, "little")
    with pytest.raises(H.HostError) as e:
        H.aggregate(hp, hdk, f["instances"], bytes(cut + prb[4 + ln0:]), f["n"], H.MOS_BDFG21, H.TRANSCRIPT_POSEIDON_DEVICE)
    assert e.value.code == H.ERR_TRANSCRIPT
    hp.close()
    hdk.close()


@pytest.mark.parametrize("teams", ["1", "3"])
def test_decide_all_1024_distinct_accumulators_with_invalid_ones_at_known_indices(gpu_ctx, fx, teams, monkeypatch):
    """`decide_all` over 1 024 DISTINCT valid accumulators, k of them replaced by invalid ones at known indices:
    per-accumulator verdicts exactly as expected, for both forms of the decide kernel, through the context API and
    the host mirror; a sample cross-checked with the C oracle's pairing."""
    import snark_verifier_amd as sv

    monkeypatch.setenv("SNARKV_DECIDE_FORM", teams)
    dkb = fx["dk"]
    accs = bytearray(fx["accs"])
    bad_idx = [0, 1, 63, 64, 511, 777, 1000, 1023]
    g = O.g1_to_bytes(O.G1_GEN)
    for k, i in enumerate(bad_idx):
        a = bytes(accs[128 * i:128 * i + 128])
        if k % 3 == 0:      # lhs + G
            a = C.g1_add(a[:64], g) + a[64:]
        elif k % 3 == 1:    # lhs and rhs swapped
            a = a[64:] + a[:64]
        else:               # rhs doubled
            a = a[:64] + C.g1_add(a[64:], a[64:])
        accs[128 * i:128 * i + 128] = a
    accs = bytes(accs)
    expected = [i not in bad_idx for i in range(1024)]
    dk = sv.DecidingKey(gpu_ctx, dkb[:64], dkb[64:192], dkb[192:320])
    allok, oks = gpu_ctx.decide_batch(dk, accs)
    assert not allok and oks == expected
    allok, oks = gpu_ctx.decide_batch(dk, fx["accs"])
    assert allok and all(oks)
    for m in (1, 2, 3, 64, 65, 512, 513):  # ragged batch sizes around the kernel-form switch (<= 512 accumulators)
        allok, oks = gpu_ctx.decide_batch(dk, accs[:128 * m])
        assert oks == expected[:m] and allok == all(expected[:m])
    dk.close()
    hdk = H.DecidingKey(dkb)
    allok, oks = H.kzg_decide_all(hdk, accs)
    assert not allok and oks == expected
    hdk.close()
    # C oracle on a sample including every bad index (a CPU decide costs ~2 ms)
    sample = sorted(set(bad_idx + list(range(0, 1024, 37))))
    sub = b"".join(accs[128 * i:128 * i + 128] for i in sample)
    _, coks = C.kzg_decide_all(dkb[64:192], dkb[192:320], sub, os.cpu_count() or 1)
    assert coks == [expected[i] for i in sample]


def test_snark_in_the_reference_serialisation_end_to_end(fx, handles):
    """N3: a (protocol, instances, proof) triple written as the SDK's bincode `Snark` (snark-verifier-sdk/src/lib.rs:47-53,
    both field encodings) -> snarkv_host_snark_parse -> PlonkVerifier::verify on the device: accept; one flipped
    proof byte: reject."""
    import random
    import struct

    import interchange_fmt as X
    import plonk_synth as S

    rng = random.Random(0xC5C5)  # gen_bench_proofs.main_distinct's protocol
    pr, _ = S.standard_plonk_protocol(rng)
    assert S.pack_protocol(pr) == fx["protocol"]
    # proof 5 and its instances out of the packed streams
    ib, off = fx["instances"], 0
    for _ in range(5):
        cols, = struct.unpack_from("<I", ib, off)
        off += 4
        for _ in range(cols):
            m, = struct.unpack_from("<I", ib, off)
            off += 4 + 32 * m
    cols, = struct.unpack_from("<I", ib, off)
    o2, inst = off + 4, []
    for _ in range(cols):
        m, = struct.unpack_from("<I", ib, o2)
        inst.append([int.from_bytes(ib[o2 + 4 + 32 * j:o2 + 36 + 32 * j], "little") for j in range(m)])
        o2 += 4 + 32 * m
    prb, off = fx["proofs"], 0
    for _ in range(5):
        off += 4 + int.from_bytes(prb[off:off + 4], "little")
    proof = prb[off + 4:off + 4 + int.from_bytes(prb[off:off + 4], "little")]
    hdk = handles[1]
    for mode in ("canonical", "montgomery"):
        s = H.Snark(X.snark_to_bincode(pr, inst, proof, mode), H.PROTOCOL_BINCODE)
        assert H.plonk_verify(s.protocol, hdk, s.instances, H.pack_proofs([s.proof]), 1)
        accs = H.plonk_succinct_verify_batch(s.protocol, hdk, s.instances, H.pack_proofs([s.proof]), 1)
        assert accs == fx["accs"][128 * 5:128 * 6]
        badp = bytearray(s.proof)
        badp[-100] ^= 1
        try:
            assert not H.plonk_verify(s.protocol, hdk, s.instances, H.pack_proofs([bytes(badp)]), 1)
        except H.HostError as e:
            assert e.code == H.ERR_TRANSCRIPT
        s.close()
