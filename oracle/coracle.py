"""ctypes wrapper of oracle/liboracle_bn254.so -- TEST INFRASTRUCTURE ONLY.
(C restatement of the reference's CPU algorithms; see c/bn254_oracle.c.)"""
import ctypes
import os
import subprocess

HERE = os.path.dirname(os.path.abspath(__file__))
_PATH = os.path.join(HERE, "liboracle_bn254.so")
_lib = None


def build():
    subprocess.run(["make", "-s", "-C", HERE], check=True)
    return _PATH


def lib():
    global _lib
    if _lib is None:
        if not os.path.exists(_PATH):
            build()
        L = ctypes.CDLL(_PATH)
        c, z, vp = ctypes.c_char_p, ctypes.c_size_t, ctypes.c_void_p
        L.oracle_g1_msm_naive.argtypes = [c, c, z, vp]
        L.oracle_g1_msm_batched.argtypes = [c, c, vp, z, vp]
        L.oracle_g1_msm_pippenger.argtypes = [c, c, z, ctypes.c_int, vp]
        L.oracle_g1_add.argtypes = [c, c, vp]
        L.oracle_g1_mul.argtypes = [c, c, vp]
        L.oracle_g1_is_on_curve.argtypes = [c]
        L.oracle_sample_scalars.argtypes = [ctypes.c_uint64, z, z, vp]
        L.oracle_sample_scalars.restype = None
        L.oracle_sample_points.argtypes = [ctypes.c_uint64, z, z, vp]
        L.oracle_sample_points.restype = None
        L.oracle_kzg_decide.argtypes = [c, c, c]
        L.oracle_kzg_pairing_value.argtypes = [c, c, c, vp]
        L.oracle_kzg_decide_all.argtypes = [c, c, c, z, ctypes.c_int, vp]
        L.oracle_selftest_cyclotomic.argtypes = [c, c]
        _lib = L
    return _lib


def msm_naive(scalars, points):
    n = len(scalars) // 32
    out = ctypes.create_string_buffer(64)
    if lib().oracle_g1_msm_naive(scalars, points, n, out) != 0:
        raise ValueError("empty MSM (reference panics: native.rs:69)")
    return out.raw


def msm_batched(scalars, points, offsets):
    import array

    offs = array.array("I", offsets)
    n_msm = len(offs) - 1
    out = ctypes.create_string_buffer(64 * n_msm)
    addr, _ = offs.buffer_info()
    if lib().oracle_g1_msm_batched(scalars, points, ctypes.c_void_p(addr), n_msm, out) != 0:
        raise ValueError("empty MSM segment")
    return out.raw


def msm_pippenger(scalars, points, threads=1):
    n = len(scalars) // 32
    out = ctypes.create_string_buffer(64)
    if lib().oracle_g1_msm_pippenger(scalars, points, n, threads, out) != 0:
        raise ValueError("empty MSM (reference panics: msm.rs:265)")
    return out.raw


def g1_add(p, q):
    out = ctypes.create_string_buffer(64)
    lib().oracle_g1_add(p, q, out)
    return out.raw


def g1_mul(p, k):
    out = ctypes.create_string_buffer(64)
    lib().oracle_g1_mul(p, k, out)
    return out.raw


def g1_is_on_curve(p):
    return bool(lib().oracle_g1_is_on_curve(p))


def sample_scalars(seed, n, first=0):
    out = ctypes.create_string_buffer(32 * n)
    lib().oracle_sample_scalars(seed, first, n, out)
    return out.raw


def sample_points(seed, n, first=0):
    out = ctypes.create_string_buffer(64 * n)
    lib().oracle_sample_points(seed, first, n, out)
    return out.raw


def kzg_decide(g2, s_g2, acc):
    """`KzgAs::decide` (decider.rs:70-82): e(lhs, g2) e(rhs, -s_g2) == 1.  g2 / s_g2: 128 B, acc = lhs|rhs: 128 B."""
    return bool(lib().oracle_kzg_decide(g2, s_g2, acc))


def kzg_pairing_value(g2, s_g2, acc):
    """The Gt element itself: 12 x 32 B in tower order (= oracle/bn254.py Fq12.to_bytes)."""
    out = ctypes.create_string_buffer(384)
    lib().oracle_kzg_pairing_value(g2, s_g2, acc, out)
    return out.raw


def kzg_decide_all(g2, s_g2, accs, threads=1):
    """`decide_all` (decider.rs:84-93): (all accepted, [per-accumulator verdicts])."""
    m = len(accs) // 128
    ok = ctypes.create_string_buffer(max(m, 1))
    allok = lib().oracle_kzg_decide_all(g2, s_g2, accs, m, threads, ok)
    return bool(allok), [b != 0 for b in ok.raw[:m]]
