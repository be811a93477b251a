"""ctypes binding of include/snarkv_host.h -- the C API of the C++ host mirror (libsnarkv_host.so).

Everything here moves bytes; the verifier logic is the C++ mirror of the reference's API
(snark-verifier_amd/host/*.hpp) and every EC operation runs on the device behind it.
`tests/test_capi_symbols.py` checks that `_SIGNATURES` and the header agree."""
import ctypes
import os

HERE = os.path.dirname(os.path.abspath(__file__))
_LIB_NAME = "libsnarkv_host.so"

MOS_GWC19, MOS_BDFG21 = 0, 1
TRANSCRIPT_EVM, TRANSCRIPT_POSEIDON, TRANSCRIPT_POSEIDON_DEVICE, TRANSCRIPT_POSEIDON_AUTO = 0, 1, 2, 3
PROTOCOL_PACKED, PROTOCOL_SERDE_JSON, PROTOCOL_BINCODE = 0, 1, 2
ERR_TRANSCRIPT, ERR_INVALID_INSTANCES, ERR_INVALID_PROTOCOL, ERR_OTHER, ERR_TRAILING = -10, -11, -12, -13, -14
ERR_CAPACITY, ERR_ARG, ERR_PANIC, ERR_DEVICE = -6, -5, -100, -101

_vp, _cp, _sz, _u32, _int = ctypes.c_void_p, ctypes.c_char_p, ctypes.c_size_t, ctypes.c_uint32, ctypes.c_int
_pp = ctypes.POINTER(ctypes.c_void_p)
_psz = ctypes.POINTER(ctypes.c_size_t)

# name -> (restype, argtypes); every function include/snarkv_host.h declares
_SIGNATURES = {
    "snarkv_host_last_error": (_cp, []),
    "snarkv_host_protocol_parse": (_int, [_cp, _sz, _int, _pp]),
    "snarkv_host_protocol_free": (None, [_vp]),
    "snarkv_host_protocol_pack": (_int, [_vp, _vp, _sz, _psz]),
    "snarkv_host_snark_parse": (_int, [_cp, _sz, _int, _pp]),
    "snarkv_host_snark_free": (None, [_vp]),
    "snarkv_host_snark_protocol": (_vp, [_vp]),
    "snarkv_host_snark_instances": (_int, [_vp, _vp, _sz, _psz]),
    "snarkv_host_snark_proof": (_int, [_vp, _vp, _sz, _psz]),
    "snarkv_host_dk_create": (_int, [_cp, _cp, _cp, _pp]),
    "snarkv_host_dk_free": (None, [_vp]),
    "snarkv_host_plonk_succinct_verify_batch": (_int, [_vp, _vp, _int, _int, _cp, _sz, _cp, _sz, _u32, _int, _vp, _sz,
                                                        ctypes.POINTER(_u32)]),
    "snarkv_host_kzg_as_accumulate": (_int, [_cp, _u32, _vp, _vp]),
    "snarkv_host_kzg_as_create_proof": (_int, [_cp, _u32, _int, _cp, _cp, _vp, _vp, _sz, _psz, _vp]),
    "snarkv_host_kzg_as_verify": (_int, [_cp, _u32, _int, _int, _cp, _sz, _vp, _vp]),
    "snarkv_host_kzg_decide": (_int, [_vp, _cp]),
    "snarkv_host_kzg_decide_all": (_int, [_vp, _cp, _u32, _vp]),
    "snarkv_host_aggregate": (_int, [_vp, _vp, _int, _int, _cp, _sz, _cp, _sz, _u32, ctypes.c_uint,
                                     ctypes.POINTER(ctypes.c_double), _vp]),
    "snarkv_host_aggregate_many": (_int, [_vp, _vp, _int, _int, _cp, _sz, _cp, _sz, ctypes.POINTER(ctypes.c_uint32), _u32,
                                          ctypes.c_uint, ctypes.POINTER(ctypes.c_double), _vp, _vp]),
    "snarkv_host_plonk_verify": (_int, [_vp, _vp, _int, _int, _cp, _sz, _cp, _sz, _u32]),
    "snarkv_host_accumulator_to_limbs": (_int, [_cp, _vp]),
    "snarkv_host_accumulator_from_limbs": (_int, [_cp, _vp]),
}

_lib = None


class HostError(RuntimeError):
    def __init__(self, code, detail=""):
        self.code = code
        super().__init__("snarkv_host error %d: %s" % (code, detail))


def lib_path():
    return os.environ.get("SNARKV_HOST_LIB") or os.path.join(HERE, _LIB_NAME)


def load_library():
    """Loads libsnarkv_host.so (which binds libsnarkv_amd.so next to it).  No fallback: a missing library raises."""
    global _lib
    if _lib is None:
        path = lib_path()
        if not os.path.exists(path):
            raise HostError(ERR_DEVICE, "%s not built (run `python __graft_entry__.py`)" % path)
        # the device library first, through its own loader: it brings torch's HIP runtime in before anything binds
        # to /opt/rocm's copy (one HIP runtime per process; see _lib.load_library)
        from ._lib import load_library as _load_device_library

        _load_device_library()
        L = ctypes.CDLL(path)
        for name, (res, args) in _SIGNATURES.items():
            fn = getattr(L, name)
            fn.restype, fn.argtypes = res, args
        _lib = L
    return _lib


def _check(rc):
    """negative codes raise; 1 / 0 (accept / reject) pass through"""
    if rc < 0:
        raise HostError(rc, (load_library().snarkv_host_last_error() or b"").decode(errors="replace"))
    return rc


def _grow(call):
    n = ctypes.c_size_t(0)
    call(None, 0, ctypes.byref(n))
    buf = ctypes.create_string_buffer(max(1, n.value))
    _check(call(buf, len(buf), ctypes.byref(n)))
    return buf.raw[: n.value]


class Protocol:
    """A parsed `PlonkProtocol` (verifier/plonk/protocol.rs:19-71)."""

    def __init__(self, data, fmt=PROTOCOL_PACKED, _borrowed=None, _owner=None):
        self._L = load_library()
        self._owner = _owner
        if _borrowed is not None:
            self._h, self._own = _borrowed, False
            return
        h = ctypes.c_void_p()
        _check(self._L.snarkv_host_protocol_parse(bytes(data), len(data), fmt, ctypes.byref(h)))
        self._h, self._own = h, True

    def pack(self):
        return _grow(lambda out, cap, n: self._L.snarkv_host_protocol_pack(self._h, out, cap, n))

    def close(self):
        if self._own and self._h:
            self._L.snarkv_host_protocol_free(self._h)
            self._h = None

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass


class Snark:
    """The SDK's `Snark { protocol, instances, proof }` (snark-verifier-sdk/src/lib.rs:47-53) from its bincode
    or serde_json serialisation."""

    def __init__(self, data, fmt=PROTOCOL_BINCODE):
        self._L = load_library()
        h = ctypes.c_void_p()
        _check(self._L.snarkv_host_snark_parse(bytes(data), len(data), fmt, ctypes.byref(h)))
        self._h = h
        self.protocol = Protocol(None, _borrowed=ctypes.c_void_p(self._L.snarkv_host_snark_protocol(h)), _owner=self)
        self.instances = _grow(lambda out, cap, n: self._L.snarkv_host_snark_instances(h, out, cap, n))
        self.proof = _grow(lambda out, cap, n: self._L.snarkv_host_snark_proof(h, out, cap, n))

    def close(self):
        if self._h:
            self._L.snarkv_host_snark_free(self._h)
            self._h = None

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass


class DecidingKey:
    """`KzgDecidingKey` (pcs/kzg/decider.rs:6-42): g1 (64 B), g2 (128 B), s_g2 (128 B); or the 320-byte concatenation."""

    def __init__(self, g1, g2=None, s_g2=None):
        self._L = load_library()
        if g2 is None:
            g1, g2, s_g2 = g1[:64], g1[64:192], g1[192:320]
        assert len(g1) == 64 and len(g2) == 128 and len(s_g2) == 128
        h = ctypes.c_void_p()
        _check(self._L.snarkv_host_dk_create(bytes(g1), bytes(g2), bytes(s_g2), ctypes.byref(h)))
        self._h = h

    def close(self):
        if self._h:
            self._L.snarkv_host_dk_free(self._h)
            self._h = None

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass


def pack_proofs(proofs):
    return b"".join(len(p).to_bytes(4, "little") + bytes(p) for p in proofs)


def plonk_succinct_verify_batch(protocol, dk, instances, proofs, n, mos=MOS_GWC19, transcript=TRANSCRIPT_EVM, strict=False):
    """N x PlonkSuccinctVerifier::verify in one device launch -> every accumulator (128 B each, proof order).
    `instances`: the n packed instance blocks concatenated; `proofs`: n x (u32 len | bytes) concatenated."""
    L = load_library()
    cap = 128 * 8 * max(1, n)
    while True:
        out = ctypes.create_string_buffer(cap)
        cnt = _u32(0)
        rc = L.snarkv_host_plonk_succinct_verify_batch(protocol._h, dk._h, mos, transcript, instances, len(instances), proofs,
                                                       len(proofs), n, 1 if strict else 0, out, cap, ctypes.byref(cnt))
        if rc == ERR_CAPACITY:
            cap = 128 * max(cnt.value, 2 * cap // 128)
            continue
        _check(rc)
        if rc != 1:  # 0 = Error::AssertionFailure mapped to "reject" by the C API: never a successful shard of zero accumulators
            raise HostError(0, "plonk_succinct_verify_batch: verifier rejected (%s)" % (load_library().snarkv_host_last_error() or b"").decode(errors="replace"))
        return out.raw[: 128 * cnt.value]


def kzg_as_accumulate(accs):
    """KzgAs::create_proof (non-zk, fresh EVM transcript) over the accumulators -> (accumulator 128 B, challenge r 32 B)"""
    L = load_library()
    acc, r = ctypes.create_string_buffer(128), ctypes.create_string_buffer(32)
    _check(L.snarkv_host_kzg_as_accumulate(bytes(accs), len(accs) // 128, acc, r))
    return acc.raw, r.raw


def kzg_as_create_proof(accs, transcript=TRANSCRIPT_EVM, pk=None, blind_scalar=None):
    """`KzgAs::create_proof` in full (accumulation.rs:148-197): `pk` = (g | s_g) 128 B and `blind_scalar` 32 B select the zk
    branch.  -> (accumulator 128 B, proof bytes, challenge r 32 B)"""
    L = load_library()
    acc, r = ctypes.create_string_buffer(128), ctypes.create_string_buffer(32)
    proof, n = ctypes.create_string_buffer(256), ctypes.c_size_t(0)
    _check(L.snarkv_host_kzg_as_create_proof(bytes(accs), len(accs) // 128, transcript, pk, blind_scalar, acc, proof, len(proof),
                                             ctypes.byref(n), r))
    return acc.raw, proof.raw[: n.value], r.raw


def kzg_as_verify(accs, proof, transcript=TRANSCRIPT_EVM, zk=False):
    """`KzgAsProof::read` + `KzgAs::verify` (accumulation.rs:114-137, 41-63) on caller-supplied proof bytes
    -> (accumulator 128 B, challenge r 32 B)"""
    L = load_library()
    acc, r = ctypes.create_string_buffer(128), ctypes.create_string_buffer(32)
    _check(L.snarkv_host_kzg_as_verify(bytes(accs), len(accs) // 128, transcript, 1 if zk else 0, bytes(proof), len(proof), acc, r))
    return acc.raw, r.raw


def kzg_decide(dk, acc):
    return _check(load_library().snarkv_host_kzg_decide(dk._h, bytes(acc))) == 1


def kzg_decide_all(dk, accs):
    """(all accepted, per-accumulator verdicts)"""
    m = len(accs) // 128
    ok = ctypes.create_string_buffer(max(1, m))
    rc = _check(load_library().snarkv_host_kzg_decide_all(dk._h, bytes(accs), m, ok))
    return rc == 1, [b != 0 for b in ok.raw[:m]]


def aggregate(protocol, dk, instances, proofs, n, mos=MOS_GWC19, transcript=TRANSCRIPT_EVM, host_threads=0, timings=False):
    """succinct-verify n proofs, accumulate, decide -> (accepted, accumulator 128 B[, timings dict])"""
    L = load_library()
    tm = (ctypes.c_double * 6)()
    acc = ctypes.create_string_buffer(128)
    rc = _check(L.snarkv_host_aggregate(protocol._h, dk._h, mos, transcript, instances, len(instances), proofs, len(proofs), n,
                                        host_threads, tm, acc))
    if timings:
        names = ("read_proofs", "fr_algebra", "msm_device", "accumulate", "decide", "total")
        return rc == 1, acc.raw, dict(zip(names, list(tm)))
    return rc == 1, acc.raw


def aggregate_many(protocol, dk, instances, proofs, job_sizes, mos=MOS_GWC19, transcript=TRANSCRIPT_EVM, host_threads=0,
                   timings=False):
    """several aggregation jobs in one call (`snarkv_host_aggregate_many`): `job_sizes[k]` consecutive proofs of the blobs
    are job k -> (every job accepted, [accumulator 128 B per job], [verdict per job][, timings dict])"""
    L = load_library()
    tm = (ctypes.c_double * 6)()
    J = len(job_sizes)
    sizes = (ctypes.c_uint32 * J)(*job_sizes)
    accs = ctypes.create_string_buffer(128 * J)
    ok = ctypes.create_string_buffer(J)
    rc = _check(L.snarkv_host_aggregate_many(protocol._h, dk._h, mos, transcript, instances, len(instances), proofs, len(proofs),
                                             sizes, J, host_threads, tm, accs, ok))
    out = (rc == 1, [accs.raw[128 * k:128 * k + 128] for k in range(J)], [b != 0 for b in ok.raw[:J]])
    if timings:
        names = ("read_proofs", "fr_algebra", "msm_device", "accumulate", "decide", "total")
        return out + (dict(zip(names, list(tm))),)
    return out


def plonk_verify(protocol, dk, instances, proofs, n, mos=MOS_GWC19, transcript=TRANSCRIPT_EVM):
    L = load_library()
    return _check(L.snarkv_host_plonk_verify(protocol._h, dk._h, mos, transcript, instances, len(instances), proofs, len(proofs), n)) == 1


def accumulator_to_limbs(acc):
    out = ctypes.create_string_buffer(512)
    _check(load_library().snarkv_host_accumulator_to_limbs(bytes(acc), out))
    return out.raw


def accumulator_from_limbs(limbs):
    out = ctypes.create_string_buffer(128)
    _check(load_library().snarkv_host_accumulator_from_limbs(bytes(limbs), out))
    return out.raw


def read_fixture(path):
    """tests/golden/bench_plonk_*.bin (layout in tests/golden/gen_bench_proofs.py): data only.
    -> dict(n, protocol, instances, proofs, dk, expected_acc, accs or None)"""
    import struct

    b = open(path, "rb").read()
    assert b[:4] == b"SVB1"
    n, = struct.unpack_from("<I", b, 4)
    off, parts = 8, []
    for _ in range(3):
        ln, = struct.unpack_from("<I", b, off)
        parts.append(b[off + 4:off + 4 + ln])
        off += 4 + ln
    dk, exp = b[off:off + 320], b[off + 320:off + 448]
    off += 448
    accs = None
    if off < len(b):
        m, = struct.unpack_from("<I", b, off)
        accs = b[off + 4:off + 4 + 128 * m]
        assert len(accs) == 128 * m
    return {"n": n, "protocol": parts[0], "instances": parts[1], "proofs": parts[2], "dk": dk, "expected_acc": exp, "accs": accs}
