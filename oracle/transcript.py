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
