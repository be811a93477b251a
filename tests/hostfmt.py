"""Byte packing for snark-verifier_amd/host/test_driver.cpp scenario inputs."""
import ctypes
import os
import struct

import bn254 as O

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def use_curve(mod):
    """Pack scenario inputs for another curve module (oracle/pallas.py): scalars reduce mod ITS r."""
    global O
    O = mod


def load_host_lib():
    import snark_verifier_amd as sv

    sv.load_library()  # brings in torch's HIP runtime first, then libsnarkv_amd.so
    import importlib.util

    spec = importlib.util.spec_from_file_location("_snarkv_build", os.path.join(ROOT, "snark-verifier_amd", "build.py"))
    b = importlib.util.module_from_spec(spec)
    spec.loader.exec_module(b)
    b.build_host_driver()  # g++ only; no-op unless a host header is newer than the library
    return ctypes.CDLL(os.path.join(ROOT, "snark-verifier_amd", "libsnarkv_hosttest.so"))


def fr(x):
    return O.fe_to_bytes(x % O.R)


def g1(p):
    return O.g1_to_bytes(p)


def u32(x):
    return struct.pack("<I", x)


def pack_msm(m):
    out = u32(1 if m.constant is not None else 0)
    if m.constant is not None:
        out += fr(m.constant)
    out += u32(len(m.scalars))
    for s, b in zip(m.scalars, m.bases):
        out += fr(s) + g1(b)
    return out


def pack_commitments(ms):
    return u32(len(ms)) + b"".join(pack_msm(m) for m in ms)


def pack_queries(qs):
    return u32(len(qs)) + b"".join(u32(p) + fr(s) + fr(e) for p, s, e in qs)


def pack_gwc19(inst):
    return (g1(inst["g"]) + pack_commitments(inst["commitments"]) + fr(inst["z"]) + pack_queries(inst["queries"])
            + fr(inst["v"]) + u32(len(inst["ws"])) + b"".join(g1(w) for w in inst["ws"]) + fr(inst["u"]))


def pack_bdfg21(inst):
    return (g1(inst["g"]) + pack_commitments(inst["commitments"]) + fr(inst["z"]) + pack_queries(inst["queries"])
            + fr(inst["mu"]) + fr(inst["gamma"]) + g1(inst["w"]) + fr(inst["z_prime"]) + g1(inst["w_prime"]))


def msm_expected(m, gen):
    """oracle value of Msm::evaluate via the C oracle (fast), same pair order"""
    import coracle as C

    prs = m.pairs(gen)
    return C.msm_naive(b"".join(fr(s) for s, _ in prs), b"".join(g1(b) for _, b in prs))
