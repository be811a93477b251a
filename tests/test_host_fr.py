"""CPU: the host mirror's Fr (snark-verifier_amd/host/fr.hpp) vs big integers."""
import ctypes
import random

import bn254 as O


def test_fr_mul_inv():
    from hostfmt import load_host_lib

    L = load_host_lib()
    rng = random.Random(3)
    o = ctypes.create_string_buffer(32)
    vals = [0, 1, 2, O.R - 1, O.R - 2] + [rng.randrange(O.R) for _ in range(200)]
    for a in vals:
        for b in vals[:8] + [rng.choice(vals)]:
            L.hd_fr_mul(O.fe_to_bytes(a), O.fe_to_bytes(b), o)
            assert O.fe_from_bytes(o.raw) == a * b % O.R
    # `invert` is Kaliski's almost-inverse + a 2^k correction whose branch depends on the iteration count k:
    # small values, powers of two (few subtractions, many shifts), r - small and random values cover both branches
    edge = [1 << i for i in range(0, 254, 7)] + [(1 << i) - 1 for i in range(2, 254, 11)] + [O.R - (1 << i) for i in range(0, 250, 13)]
    # the branch is taken on the MONTGOMERY residue x = a * 2^256: small and power-of-two x (k < 257, and the k = 257 / 258
    # cases whose unreduced operand meets mont_mul) need a = x * 2^-256
    rinv = pow(1 << 256, -1, O.R)
    edge += [(1 << j) * rinv % O.R for j in list(range(0, 12)) + [64, 128, 200, 252, 253]] + [(O.R - 1) * rinv % O.R, 3 * rinv % O.R]
    o2 = ctypes.create_string_buffer(32)
    for a in vals[1:] + edge + [rng.randrange(1, 1 << rng.randrange(1, 254)) for _ in range(300)]:
        assert L.hd_fr_inv(O.fe_to_bytes(a), o) == 1
        assert O.fe_from_bytes(o.raw) == pow(a, -1, O.R)
        assert L.hd_fr_inv_fermat(O.fe_to_bytes(a), o2) == 1 and o2.raw == o.raw
    assert L.hd_fr_inv(O.fe_to_bytes(0), o) == 0


def test_host_pool_selftest():
    """The host pool (host/loader.hpp `HostPool`, round 4: lock-free joining, spinning workers, the caller works too):
    every item exactly once over many (n, threads, grain) shapes, nested fan-out inline, exceptions reach the caller and the
    pool survives them, several submitters take turns, a pool task is refused the device lock."""
    from hostfmt import load_host_lib

    L = load_host_lib()
    L.hd_pool_selftest.argtypes = [ctypes.c_int]
    assert L.hd_pool_selftest(600) == 0
