"""Oracle (TEST INFRASTRUCTURE ONLY): the reference's Keccak/EVM transcript on the
native loader, restated in Python.

Follows snark-verifier/src/system/halo2/transcript/evm.rs:175-268 (squeeze /
common / read) and :373-398 (write); `u256_to_fe` is loader/evm/util.rs:61-67.
Keccak-256 is the original Keccak padding (0x01), i.e. the `sha3::Keccak256`
the reference imports (evm.rs:16, external crate `sha3` 0.10) -- restated from
FIPS-202 / the Keccak reference, and pinned by (a) the permutation reproducing
hashlib's SHA3-256 when run with the 0x06 padding and (b) the public Keccak-256
vectors of "" and "abc" (tests/test_transcript.py).

PARITY UNPINNED against the Rust crate itself (cannot be built here); the
byte-level behaviour is fully determined by the cited lines + Keccak-256.
"""
import bn254 as O

_RC = [
    0x0000000000000001, 0x0000000000008082, 0x800000000000808A, 0x8000000080008000,
    0x000000000000808B, 0x0000000080000001, 0x8000000080008081, 0x8000000000008009,
    0x000000000000008A, 0x0000000000000088, 0x0000000080008009, 0x000000008000000A,
    0x000000008000808B, 0x800000000000008B, 0x8000000000008089, 0x8000000000008003,
    0x8000000000008002, 0x8000000000000080, 0x000000000000800A, 0x800000008000000A,
    0x8000000080008081, 0x8000000000008080, 0x0000000080000001, 0x8000000080008008,
]
_M = (1 << 64) - 1


def _rotl(x, n):
    n %= 64
    return ((x << n) | (x >> (64 - n))) & _M if n else x


def keccak_f1600(a):
    """a: 25 lanes, index x + 5y.  In place; textbook theta/rho/pi/chi/iota."""
    for rnd in range(24):
        c = [a[x] ^ a[x + 5] ^ a[x + 10] ^ a[x + 15] ^ a[x + 20] for x in range(5)]
        d = [c[(x - 1) % 5] ^ _rotl(c[(x + 1) % 5], 1) for x in range(5)]
        for i in range(25):
            a[i] ^= d[i % 5]
        # rho + pi: B[y, 2x+3y] = rot(A[x, y], r[x, y])
        b = [0] * 25
        x, y = 1, 0
        b[0] = a[0]
        for t in range(24):
            r = ((t + 1) * (t + 2) // 2) % 64
            nx, ny = y, (2 * x + 3 * y) % 5
            b[nx + 5 * ny] = _rotl(a[x + 5 * y], r)
            x, y = nx, ny
        for yy in range(5):
            for xx in range(5):
                a[xx + 5 * yy] = b[xx + 5 * yy] ^ ((~b[(xx + 1) % 5 + 5 * yy]) & _M & b[(xx + 2) % 5 + 5 * yy])
        a[0] ^= _RC[rnd]
    return a


def _sponge256(data, pad_byte):
    rate = 136
    st = [0] * 25
    msg = bytearray(data)
    msg.append(pad_byte)
    while len(msg) % rate:
        msg.append(0)
    msg[-1] |= 0x80
    for off in range(0, len(msg), rate):
        for i in range(rate // 8):
            st[i] ^= int.from_bytes(msg[off + 8 * i:off + 8 * i + 8], "little")
        keccak_f1600(st)
    return b"".join(st[i].to_bytes(8, "little") for i in range(4))


def keccak256(data):
    return _sponge256(data, 0x01)


def sha3_256(data):  # only to pin the permutation against hashlib
    return _sponge256(data, 0x06)


class TranscriptError(Exception):
    """`Error::Transcript(kind, msg)`"""


class EvmTranscript:
    """evm.rs:134-268 + 373-398 on the native loader; `stream` is the proof bytes."""

    def __init__(self, stream=b""):
        self.stream = bytearray(stream)
        self.pos = 0
        self.buf = bytearray()

    # evm.rs:184-198
    def squeeze_challenge(self):
        data = bytes(self.buf) + (b"\x01" if len(self.buf) == 0x20 else b"")
        h = keccak256(data)
        self.buf = bytearray(h)
        return int.from_bytes(h, "big") % O.R  # u256_to_fe

    # evm.rs:200-216
    def common_ec_point(self, pt):
        if pt is None:
            raise TranscriptError("Invalid elliptic curve point")
        self.buf += pt[0].to_bytes(32, "big") + pt[1].to_bytes(32, "big")

    # evm.rs:218-222
    def common_scalar(self, s):
        self.buf += s.to_bytes(32, "big")

    def _read(self, n):
        if self.pos + n > len(self.stream):
            raise TranscriptError("failed to fill whole buffer")
        b = bytes(self.stream[self.pos:self.pos + n])
        self.pos += n
        return b

    # evm.rs:231-245
    def read_scalar(self):
        v = int.from_bytes(self._read(32), "big")
        if v >= O.R:
            raise TranscriptError("Invalid scalar encoding in proof")
        self.common_scalar(v)
        return v

    # evm.rs:247-268
    def read_ec_point(self):
        x = int.from_bytes(self._read(32), "big")
        y = int.from_bytes(self._read(32), "big")
        if x >= O.P or y >= O.P or not O.g1_is_on_curve((x, y)):
            raise TranscriptError("Invalid elliptic curve point encoding in proof")
        self.common_ec_point((x, y))
        return (x, y)

    # evm.rs:373-388
    def write_ec_point(self, pt):
        self.common_ec_point(pt)
        self.stream += pt[0].to_bytes(32, "big") + pt[1].to_bytes(32, "big")

    # evm.rs:390-396
    def write_scalar(self, s):
        self.common_scalar(s)
        self.stream += s.to_bytes(32, "big")

    def finalize(self):
        return bytes(self.stream)


# ---------------------------------------------------------------------------
# Poseidon (second half of N2).
#
# The reference's hasher (snark-verifier/src/util/hash/poseidon.rs:115-202) runs
# the OPTIMISED schedule (pre-added constants, sparse MDS) whose tables come from
# the un-vendored crate `poseidon` (privacy-scaling-explorations/poseidon,
# `Spec::new(r_f, r_p)`, poseidon.rs:7,128).  That schedule is an exact
# rewriting of the plain Poseidon permutation (Poseidon paper, appendix B), so
# this oracle states the PLAIN permutation:
#   * round constants and the Cauchy MDS from the Grain LFSR exactly as the
#     Poseidon reference (`generate_parameters_grain.sage 1 0 254 t R_F R_P`),
#   * full rounds R_F/2, partial rounds R_P, full rounds R_F/2,
# and the sponge framing of poseidon.rs:150-166 / :44-75 (inputs added to words
# 1.., a 1 added to the word after the last input, an extra permutation when
# the input length is a multiple of RATE, output = word 1).
#
# Pinned: the generator + permutation reproduce the public t = 3 parameters
# (first round constant, MDS[0][0]) and the public known answer
# poseidon([1, 2]) = 0x115cc0f5...189a of that instance (tests/test_transcript.py).
# UNPINNED (crate internals that cannot be read here): `State::default()` =
# [2^64, 0, ...] and the 32-byte compressed G1 encoding of halo2curves 0.6.0
# (bit 7 of byte 31 = identity, bit 6 = parity of y) used by read/write_ec_point.
def _grain_bits(n, t, r_f, r_p):
    bits = []

    def put(v, w):
        for i in range(w - 1, -1, -1):
            bits.append((v >> i) & 1)

    put(1, 2)    # prime field
    put(0, 4)    # S-box x^alpha
    put(n, 12)
    put(t, 12)
    put(r_f, 10)
    put(r_p, 10)
    bits.extend([1] * 30)
    st = bits

    def raw():
        nonlocal st
        nb = st[62] ^ st[51] ^ st[38] ^ st[23] ^ st[13] ^ st[0]
        st = st[1:] + [nb]
        return nb

    for _ in range(160):
        raw()
    while True:
        b1, b2 = raw(), raw()
        if b1:
            yield b2


_SPEC_CACHE = {}


def poseidon_spec(t, r_f, r_p, n_bits=254, modulus=None):
    """(round constants [(R_F+R_P)*t], MDS t x t) for the prime field `modulus` (default Fr)."""
    modulus = modulus or O.R
    key = (t, r_f, r_p, n_bits, modulus)
    if key in _SPEC_CACHE:
        return _SPEC_CACHE[key]
    g = _grain_bits(n_bits, t, r_f, r_p)

    def nbits():
        v = 0
        for _ in range(n_bits):
            v = (v << 1) | next(g)
        return v

    rc = []
    while len(rc) < (r_f + r_p) * t:
        v = nbits()
        if v < modulus:          # rejection sampling for round constants
            rc.append(v)
    while True:                   # MDS: no rejection, reduce mod p; resample until distinct
        rl = [nbits() % modulus for _ in range(2 * t)]
        if len(set(rl)) != 2 * t:
            continue
        xs, ys = rl[:t], rl[t:]
        if any((x + y) % modulus == 0 for x in xs for y in ys):
            continue
        mds = [[pow((xs[i] + ys[j]) % modulus, -1, modulus) for j in range(t)] for i in range(t)]
        break
    _SPEC_CACHE[key] = (rc, mds)
    return rc, mds


def poseidon_permute(state, r_f, r_p, modulus=None):
    modulus = modulus or O.R
    t = len(state)
    rc, mds = poseidon_spec(t, r_f, r_p, modulus=modulus)
    k = 0
    for rnd in range(r_f + r_p):
        state = [(s + rc[k + i]) % modulus for i, s in enumerate(state)]
        k += t
        if rnd < r_f // 2 or rnd >= r_f // 2 + r_p:
            state = [pow(s, 5, modulus) for s in state]
        else:
            state[0] = pow(state[0], 5, modulus)
        state = [sum(mds[i][j] * state[j] for j in range(t)) % modulus for i in range(t)]
    return state


class Poseidon:
    """poseidon.rs:115-202 (sponge framing) over the plain permutation."""

    def __init__(self, t=5, rate=4, r_f=8, r_p=60):
        self.t, self.rate, self.r_f, self.r_p = t, rate, r_f, r_p
        self.state = [1 << 64] + [0] * (t - 1)   # poseidon::State::default() (crate; unpinned)
        self.buf = []

    def update(self, elements):                   # poseidon.rs:145-147
        self.buf.extend(elements)

    def squeeze(self):                            # poseidon.rs:151-164
        buf, self.buf = self.buf, []
        exact = len(buf) % self.rate == 0
        for i in range(0, len(buf), self.rate):
            self._permutation(buf[i:i + self.rate])
        if exact:
            self._permutation([])
        return self.state[1]

    def _permutation(self, inputs):               # poseidon.rs:44-75 + :166-201
        assert len(inputs) < self.t
        for i, x in enumerate(inputs):
            self.state[1 + i] = (self.state[1 + i] + x) % O.R
        if 1 + len(inputs) < self.t:   # `.skip(1 + inputs.len())` is empty for a full-rate chunk (poseidon.rs:61-74)
            self.state[1 + len(inputs)] = (self.state[1 + len(inputs)] + 1) % O.R
        self.state = poseidon_permute(self.state, self.r_f, self.r_p)


def g1_compress(pt):
    """halo2curves 0.6.0 bn256 `G1Affine::to_bytes` as recalled (UNPINNED): x LE,
    bit 6 of byte 31 = y odd, bit 7 = identity (then x = 0)."""
    if pt is None:
        b = bytearray(32)
        b[31] |= 0x80
        return bytes(b)
    b = bytearray(pt[0].to_bytes(32, "little"))
    b[31] |= (pt[1] & 1) << 6
    return bytes(b)


def g1_decompress(b):
    """`G1Affine::from_bytes`; None result = invalid encoding (raises)."""
    b = bytearray(b)
    is_inf, ysign = b[31] >> 7, (b[31] >> 6) & 1
    b[31] &= 0x3F
    x = int.from_bytes(b, "little")
    if x >= O.P:
        raise TranscriptError("Invalid elliptic curve point encoding in proof")
    if is_inf:  # the identity has exactly one encoding: flag set, everything else zero
        if x != 0 or ysign:
            raise TranscriptError("Invalid elliptic curve point encoding in proof")
        return None
    y2 = (x * x * x + 3) % O.P
    y = pow(y2, (O.P + 1) // 4, O.P)
    if y * y % O.P != y2:
        raise TranscriptError("Invalid elliptic curve point encoding in proof")
    if (y & 1) != ysign:
        y = O.P - y
    return (x, y)


class PoseidonTranscript:
    """system/halo2/transcript/halo2.rs:170-321 on the native loader
    (T=5, RATE=4, R_F=8, R_P=60 are the reference's example parameters,
    examples/evm-verifier-with-accumulator.rs:36-39)."""

    def __init__(self, stream=b"", t=5, rate=4, r_f=8, r_p=60):
        self.stream = bytearray(stream)
        self.pos = 0
        self.buf = Poseidon(t, rate, r_f, r_p)

    def squeeze_challenge(self):                  # halo2.rs:211-213
        return self.buf.squeeze()

    def common_scalar(self, s):                   # halo2.rs:215-218
        self.buf.update([s])

    def common_ec_point(self, pt):                # halo2.rs:220-237: x, y through fe_to_fe (mod r)
        if pt is None:
            raise TranscriptError("Invalid elliptic curve point encoding in proof")
        self.buf.update([pt[0] % O.R, pt[1] % O.R])

    def _read(self, n):
        if self.pos + n > len(self.stream):
            raise TranscriptError("failed to fill whole buffer")
        b = bytes(self.stream[self.pos:self.pos + n])
        self.pos += n
        return b

    def read_scalar(self):                        # halo2.rs:247-260
        v = int.from_bytes(self._read(32), "little")
        if v >= O.R:
            raise TranscriptError("Invalid scalar encoding in proof")
        self.common_scalar(v)
        return v

    def read_ec_point(self):                      # halo2.rs:262-275
        pt = g1_decompress(self._read(32))
        self.common_ec_point(pt)                  # the identity decodes but cannot be absorbed
        return pt

    def write_scalar(self, s):                    # halo2.rs:300-309
        self.common_scalar(s)
        self.stream += s.to_bytes(32, "little")

    def write_ec_point(self, pt):                 # halo2.rs:311-320
        self.common_ec_point(pt)
        self.stream += g1_compress(pt)

    def finalize(self):
        return bytes(self.stream)


# ---------------------------------------------------------------------------
# The optimised schedule's tables (what the reference's `poseidon::Spec` holds and
# poseidon.rs:166-201 consumes), derived from the plain spec; used to feed the device
# kernel (`snarkv_poseidon_create`) in tests.  Derivation: constants slide behind the
# S-boxes (k_r = M^-1 e_{r+1}; in a partial round only the word-0 component stays, the
# rest moves in front of that round's S-box), the partial rounds' dense matrix factors
# as M~ = M'' M' with M' = diag(1, m^) pushed into the previous round.
def _mat_inv(a, mod):
    n = len(a)
    a = [row[:] + [1 if i == j else 0 for j in range(n)] for i, row in enumerate(a)]
    for c in range(n):
        p = next(r for r in range(c, n) if a[r][c] % mod)
        a[c], a[p] = a[p], a[c]
        inv = pow(a[c][c], -1, mod)
        a[c] = [x * inv % mod for x in a[c]]
        for r in range(n):
            if r != c and a[r][c]:
                f = a[r][c]
                a[r] = [(x - f * y) % mod for x, y in zip(a[r], a[c])]
    return [row[n:] for row in a]


def _mat_mul(a, b, mod):
    return [[sum(a[i][k] * b[k][j] for k in range(len(b))) % mod for j in range(len(b[0]))] for i in range(len(a))]


def poseidon_opt_tables(t, r_f, r_p, modulus=None):
    mod = modulus or O.R
    rc, mds = poseidon_spec(t, r_f, r_p, modulus=mod)
    h, R = r_f // 2, r_f + r_p
    c = lambda r: rc[r * t:(r + 1) * t]
    minv = _mat_inv(mds, mod)
    mv = lambda m, v: [sum(m[i][j] * v[j] for j in range(t)) % mod for i in range(t)]
    kfull, partial = {}, [0] * r_p
    e = c(R - 1)
    for r in range(R - 2, -1, -1):
        back = mv(minv, e)
        e = c(r)[:]
        if h <= r < h + r_p:
            partial[r - h] = back[0]
            for i in range(1, t):
                e[i] = (e[i] + back[i]) % mod
        else:
            kfull[r] = back
    start = [e] + [kfull[r] for r in range(h)]
    end = [kfull[r] for r in range(h + r_p, R - 1)]
    cur = [row[:] for row in mds]
    rows, cols = [None] * r_p, [None] * r_p
    for r in range(r_p - 1, -1, -1):
        mhat = [row[1:] for row in cur[1:]]
        mhat_inv = _mat_inv(mhat, mod)
        v = [sum(cur[0][1 + k] * mhat_inv[k][j] for k in range(t - 1)) % mod for j in range(t - 1)]
        rows[r] = [cur[0][0]] + v
        cols[r] = [cur[i][0] for i in range(1, t)]
        mprime = [[1] + [0] * (t - 1)] + [[0] + mhat[i] for i in range(t - 1)]
        cur = _mat_mul(mprime, mds, mod)
    return {"start": start, "partial": partial, "end": end, "mds": mds, "pre_sparse_mds": cur,
            "sparse_rows": rows, "sparse_col_hats": cols}


def poseidon_permute_opt(state, t, r_f, r_p, modulus=None):
    """The permutation through the optimised tables -- must equal poseidon_permute."""
    mod = modulus or O.R
    T = poseidon_opt_tables(t, r_f, r_p, modulus=mod)
    h = r_f // 2
    sb = lambda x: pow(x, 5, mod)
    mv = lambda m, v: [sum(m[i][j] * v[j] for j in range(t)) % mod for i in range(t)]
    s = [(x + k) % mod for x, k in zip(state, T["start"][0])]
    for r in range(h):
        s = [(sb(x) + k) % mod for x, k in zip(s, T["start"][r + 1])]
        s = mv(T["mds"] if r + 1 < h else T["pre_sparse_mds"], s)
    for r in range(r_p):
        s0 = (sb(s[0]) + T["partial"][r]) % mod
        new0 = (T["sparse_rows"][r][0] * s0 + sum(a * b for a, b in zip(T["sparse_rows"][r][1:], s[1:]))) % mod
        s = [new0] + [(s[i] + T["sparse_col_hats"][r][i - 1] * s0) % mod for i in range(1, t)]
    for r in range(h):
        s = [sb(x) for x in s] if r + 1 == h else [(sb(x) + k) % mod for x, k in zip(s, T["end"][r])]
        s = mv(T["mds"], s)
    return s


def poseidon_transcript_challenges(elems, seg_len, t=5, rate=4, r_f=8, r_p=60):
    """Oracle of `snarkv_poseidon_transcript_batch` for ONE transcript: absorb the segments, squeeze after each."""
    p = Poseidon(t, rate, r_f, r_p)
    out, pos = [], 0
    for n in seg_len:
        p.update(elems[pos:pos + n])
        pos += n
        out.append(p.squeeze())
    return out


# ---------------------------------------------------------------------------
# halo2's Blake2b transcript (`halo2_proofs::transcript::{Blake2bRead, Blake2bWrite}` with
# `Challenge255`), the one the reference's IPA tests use on pallas (pcs/ipa.rs:438-440,
# pcs/ipa/accumulation.rs:244-247, system/halo2/test/ipa/native.rs:11-12).  External crate, not in
# /root/reference: restated from its published definition -- BLAKE2b-512 personalised
# "Halo2-Transcript"; prefix byte 0 before a challenge, 1 before a point (x, y as 32-byte LE), 2 before
# a scalar; a challenge is the 64-byte digest of a COPY of the state, read little-endian and reduced
# mod r (`from_uniform_bytes`); points travel compressed (x with the parity of y in bit 255).
# PARITY UNPINNED.  Curve-generic: `curve` is a module like oracle/bn254.py or oracle/pallas.py.
# ---------------------------------------------------------------------------
class Blake2bTranscript:
    def __init__(self, curve, stream=b""):
        import hashlib

        self.c = curve
        self.state = hashlib.blake2b(digest_size=64, person=b"Halo2-Transcript")
        self.stream = bytearray(stream)
        self.pos = 0

    def squeeze_challenge(self):
        self.state.update(b"\x00")
        return int.from_bytes(self.state.copy().digest(), "little") % self.c.R

    def common_ec_point(self, pt):
        if pt is None:
            raise TranscriptError("cannot write points at infinity to the transcript")
        self.state.update(b"\x01" + self.c.fe_to_bytes(pt[0]) + self.c.fe_to_bytes(pt[1]))

    def common_scalar(self, s):
        self.state.update(b"\x02" + self.c.fe_to_bytes(s % self.c.R))

    def _read(self, n):
        if self.pos + n > len(self.stream):
            raise TranscriptError("failed to fill whole buffer")
        b = bytes(self.stream[self.pos:self.pos + n])
        self.pos += n
        return b

    def read_scalar(self):
        v = int.from_bytes(self._read(32), "little")
        if v >= self.c.R:
            raise TranscriptError("invalid field element encoding in proof")
        self.common_scalar(v)
        return v

    def read_ec_point(self):
        raw = int.from_bytes(self._read(32), "little")
        sign, x = raw >> 255, raw & ((1 << 255) - 1)
        y = self.c.fq_sqrt(x * x * x + self.c.B1) if x < self.c.P else None
        if y is None or (x == 0 and sign == 0):
            raise TranscriptError("invalid point encoding in proof")
        if y & 1 != sign:
            y = self.c.P - y
        self.common_ec_point((x, y))
        return (x, y)

    def write_ec_point(self, pt):
        self.common_ec_point(pt)
        self.stream += (pt[0] | ((pt[1] & 1) << 255)).to_bytes(32, "little")

    def write_scalar(self, s):
        self.common_scalar(s)
        self.stream += self.c.fe_to_bytes(s % self.c.R)

    def finalize(self):
        return bytes(self.stream)
