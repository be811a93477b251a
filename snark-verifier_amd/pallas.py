"""ctypes bindings of the pasta build (libsnarkv_pallas.so, include/snarkv_pallas.h): the large MSM
and the IPA decider on pallas -- the curve of the reference's own IPA tests.  No CPU fallback: the
loader raises if the library is missing, the calls raise `SnarkvError` on any failure."""
import ctypes
import os

from ._lib import SnarkvError, _as_bytes

HERE = os.path.dirname(os.path.abspath(__file__))
_LIB = None


def load_library():
    global _LIB
    if _LIB is None:
        from . import load_library as load_bn254

        load_bn254()  # brings in the HIP runtime torch ships, exactly as the BN254 library does
        path = os.environ.get("SNARKV_PALLAS_LIB") or os.path.join(HERE, "libsnarkv_pallas.so")
        if not os.path.exists(path):
            raise SnarkvError(-1000, "%s not built: run __graft_entry__.build()" % path)
        lib = ctypes.CDLL(path)
        vp, sz, u32 = ctypes.c_void_p, ctypes.c_size_t, ctypes.c_uint32
        for name, (res, args) in {
            "snarkv_pallas_ctx_create": (ctypes.c_int, [ctypes.c_int, vp, ctypes.POINTER(vp)]),
            "snarkv_pallas_ctx_destroy": (None, [vp]),
            "snarkv_pallas_ctx_sync": (ctypes.c_int, [vp]),
            "snarkv_pallas_ctx_wait_stream": (ctypes.c_int, [vp, vp]),
            "snarkv_pallas_stream_wait_ctx": (ctypes.c_int, [vp, vp]),
            "snarkv_pallas_ctx_stream": (vp, [vp]),
            "snarkv_pallas_ctx_host_buffer": (ctypes.c_int, [vp, ctypes.c_int, sz, ctypes.POINTER(vp)]),
            "snarkv_pallas_last_error": (ctypes.c_char_p, []),
            "snarkv_pallas_version": (ctypes.c_char_p, []),
            "snarkv_pallas_g1_msm_pippenger": (ctypes.c_int, [vp, vp, vp, sz, vp]),
            "snarkv_pallas_g1_msm_pippenger_dev": (ctypes.c_int, [vp, vp, vp, sz, ctypes.c_int, vp]),
            "snarkv_pallas_g1_msm_naive": (ctypes.c_int, [vp, vp, vp, sz, u32, vp]),
            "snarkv_pallas_g1_msm_batched": (ctypes.c_int, [vp, vp, vp, vp, sz, u32, vp]),
            "snarkv_pallas_ipa_dk_create": (ctypes.c_int, [vp, vp, sz, ctypes.POINTER(vp)]),
            "snarkv_pallas_ipa_dk_create_shard": (ctypes.c_int, [vp, vp, sz, u32, sz, ctypes.POINTER(vp)]),
            "snarkv_pallas_ipa_commit_partial_dev": (ctypes.c_int, [vp, vp, vp, vp]),
            "snarkv_pallas_ipa_dk_destroy": (None, [vp]),
            "snarkv_pallas_ipa_dk_k": (u32, [vp]),
            "snarkv_pallas_ipa_decide_batch": (ctypes.c_int, [vp, vp, vp, vp, sz, vp]),
        }.items():
            fn = getattr(lib, name)
            fn.restype, fn.argtypes = res, args
        _LIB = lib
    return _LIB


def _stream_handle(stream):
    from ._lib import stream_handle

    return stream_handle(stream)


def _check(rc):
    if rc < 0:
        raise SnarkvError(rc, (load_library().snarkv_pallas_last_error() or b"").decode())
    return rc


class PallasContext:
    """One HIP stream + scratch of the pasta library (`snarkv_pallas_ctx_*`)."""

    def __init__(self, device=0, stream=None):
        self._lib = load_library()
        self._h = ctypes.c_void_p()
        _check(self._lib.snarkv_pallas_ctx_create(device, ctypes.c_void_p(stream or 0), ctypes.byref(self._h)))

    def close(self):
        if self._h:
            self._lib.snarkv_pallas_ctx_destroy(self._h)
            self._h = ctypes.c_void_p()

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass

    def sync(self):
        _check(self._lib.snarkv_pallas_ctx_sync(self._h))

    def wait_stream(self, stream=None):
        """`snarkv_pallas_ctx_wait_stream`: the context's next work runs after everything queued on `stream` (a HIP stream
        handle or a torch stream; None = torch's current stream)."""
        _check(self._lib.snarkv_pallas_ctx_wait_stream(self._h, ctypes.c_void_p(_stream_handle(stream))))

    def stream_wait(self, stream=None):
        """`snarkv_pallas_stream_wait_ctx`: `stream`'s next work runs after everything this context has queued."""
        _check(self._lib.snarkv_pallas_stream_wait_ctx(self._h, ctypes.c_void_p(_stream_handle(stream))))

    def msm_pippenger(self, scalars, points):
        """`util::msm::multi_scalar_multiplication` on pallas (msm.rs:308-343), affine bytes."""
        s, p = _as_bytes(scalars), _as_bytes(points)
        if len(s) % 32 or len(p) % 64 or len(s) // 32 != len(p) // 64:
            raise SnarkvError(-2, "scalars/points length mismatch (reference: assert_eq!, msm.rs:309)")
        out = ctypes.create_string_buffer(64)
        _check(self._lib.snarkv_pallas_g1_msm_pippenger(self._h, s if s else b"\x00", p if p else b"\x00", len(s) // 32, out))
        return out.raw

    def msm_batched(self, scalars, points, offsets, flags=0):
        """`NativeLoader::multi_scalar_multiplication` on pallas (loader/native.rs:61-71), len(offsets)-1
        MSMs as segments of one launch -> concatenated affine bytes."""
        import array

        s, p = _as_bytes(scalars), _as_bytes(points)
        off = array.array("I", offsets)
        addr, _ = off.buffer_info()
        n_msm = len(offsets) - 1
        out = ctypes.create_string_buffer(64 * max(n_msm, 1))
        _check(self._lib.snarkv_pallas_g1_msm_batched(self._h, s if s else b"\x00", p if p else b"\x00",
                                                      ctypes.c_void_p(addr), n_msm, flags, out))
        return out.raw[:64 * n_msm]

    def msm_naive(self, scalars, points, flags=0):
        s, p = _as_bytes(scalars), _as_bytes(points)
        if len(s) % 32 or len(p) % 64 or len(s) // 32 != len(p) // 64:
            raise SnarkvError(-2, "scalars/points length mismatch")
        out = ctypes.create_string_buffer(64)
        _check(self._lib.snarkv_pallas_g1_msm_naive(self._h, s if s else b"\x00", p if p else b"\x00", len(s) // 32, flags, out))
        return out.raw

    def msm_pippenger_dev(self, d_scalars, d_points, n, d_out, window_bits=0):
        _check(self._lib.snarkv_pallas_g1_msm_pippenger_dev(self._h, d_scalars, d_points, n, window_bits, d_out))

    def ipa_dk_create(self, g):
        g = _as_bytes(g)
        h = ctypes.c_void_p()
        _check(self._lib.snarkv_pallas_ipa_dk_create(self._h, g if g else b"\x00", len(g) // 64, ctypes.byref(h)))
        return PallasIpaDecidingKey(self._lib, h)

    def ipa_decide_batch(self, dk, xi, u):
        """`IpaAs::decide_all` per accumulator on pallas (pcs/ipa/decider.rs:47-66) -> list of booleans."""
        xi, u = _as_bytes(xi), _as_bytes(u)
        m = len(u) // 64
        assert len(u) == 64 * m and len(xi) == 32 * dk.k * m
        ok = ctypes.create_string_buffer(max(m, 1))
        _check(self._lib.snarkv_pallas_ipa_decide_batch(self._h, dk._h, xi if xi else b"\x00", u if u else b"\x00", m, ok))
        return [b != 0 for b in ok.raw[:m]]


class PallasIpaDecidingKey:
    def __init__(self, lib, h):
        self._lib, self._h = lib, h
        self.k = lib.snarkv_pallas_ipa_dk_k(h)

    def close(self):
        if self._h:
            self._lib.snarkv_pallas_ipa_dk_destroy(self._h)
            self._h = ctypes.c_void_p()

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass
