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
    for a in vals[1:40]:
        assert L.hd_fr_inv(O.fe_to_bytes(a), o) == 1
        assert O.fe_from_bytes(o.raw) == pow(a, -1, O.R)
    assert L.hd_fr_inv(O.fe_to_bytes(0), o) == 0
