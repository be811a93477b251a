"""halo2curves' in-memory form of field elements (SNARKV_FLAG_MONTGOMERY): the four little-endian u64 limbs of
a * 2^256 mod r (mod p).  Pure-integer re-encoders for the tests (the oracle works on canonical bytes)."""
import bn254 as O

R256 = 1 << 256


def fe_to_mont(b32, mod):
    return (int.from_bytes(b32, "little") * R256 % mod).to_bytes(32, "little")


def fe_from_mont(b32, mod):
    return (int.from_bytes(b32, "little") * pow(R256, -1, mod) % mod).to_bytes(32, "little")


def scalars_to_mont(s):
    return b"".join(fe_to_mont(s[i:i + 32], O.R) for i in range(0, len(s), 32))


def coords_to_mont(p):
    """any run of Fq coordinates (G1: x || y, G2: x.c0 || x.c1 || y.c0 || y.c1, accumulators: lhs || rhs); the identity's
    zero bytes stay zero"""
    return b"".join(fe_to_mont(p[i:i + 32], O.P) for i in range(0, len(p), 32))


def coords_from_mont(p):
    return b"".join(fe_from_mont(p[i:i + 32], O.P) for i in range(0, len(p), 32))
