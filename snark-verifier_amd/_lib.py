"""ctypes binding of include/snarkv_amd.h.  Every symbol the header declares is
bound in `_SIGNATURES`; `tests/test_capi_symbols.py` checks the two agree."""
import ctypes
import os

HERE = os.path.dirname(os.path.abspath(__file__))
_LIB_NAME = "libsnarkv_amd.so"

SNARKV_OK = 0
SNARKV_ERR_EMPTY = -1
SNARKV_ERR_LENGTH = -2
SNARKV_ERR_ENCODING = -3
SNARKV_ERR_DEVICE = -4
SNARKV_ERR_ARG = -5
SNARKV_FLAG_VALIDATE = 1
SNARKV_FLAG_MONTGOMERY = 2  # halo2curves' in-memory form (a * 2^256 mod r / mod p) instead of the canonical wire form
SNARKV_HOST_BUFFERS = 4  # include/snarkv_amd.h
SNARKV_PIP_STAGES = 9
PIP_STAGE_NAMES = [
    "total", "prepare_glv_montgomery_histogram", "scan", "partition_sort", "bucket_accumulate",
    "bucket_combine", "bucket_reduce", "window_shift_chain", "final_to_affine",
]
G1_PARTIAL_BYTES = 144

_ERR_NAMES = {
    SNARKV_ERR_EMPTY: "EMPTY (the reference panics here: native.rs:69 / msm.rs:265)",
    SNARKV_ERR_LENGTH: "LENGTH",
    SNARKV_ERR_ENCODING: "ENCODING",
    SNARKV_ERR_DEVICE: "DEVICE",
    SNARKV_ERR_ARG: "ARG",
}


class SnarkvError(RuntimeError):
    def __init__(self, code, detail=""):
        self.code = code
        super().__init__("snarkv error %d %s %s" % (code, _ERR_NAMES.get(code, "?"), detail))


_vp = ctypes.c_void_p
_cp = ctypes.c_char_p
_sz = ctypes.c_size_t
_u32 = ctypes.c_uint32
_int = ctypes.c_int
_pp = ctypes.POINTER(ctypes.c_void_p)

# name -> (restype, argtypes); must list every function include/snarkv_amd.h declares
_SIGNATURES = {
    "snarkv_ctx_create": (_int, [_int, _vp, _pp]),
    "snarkv_ctx_destroy": (None, [_vp]),
    "snarkv_ctx_sync": (_int, [_vp]),
    "snarkv_ctx_wait_stream": (_int, [_vp, _vp]),
    "snarkv_stream_wait_ctx": (_int, [_vp, _vp]),
    "snarkv_ctx_stream": (_vp, [_vp]),
    "snarkv_ctx_host_buffer": (_int, [_vp, _int, _sz, _pp]),
    "bn254_host_buffer": (_int, [_int, _sz, _pp]),
    "snarkv_g1_decompress": (_int, [_vp, _cp, _sz, _vp, _vp]),
    "bn254_g1_decompress": (_int, [_cp, _sz, _vp, _vp]),
    "bn254_set_thread_flags": (ctypes.c_int64, [ctypes.c_int64]),
    "bn254_get_flags": (ctypes.c_uint32, []),
    "bn254_shutdown": (_int, []),
    "bn254_default_contexts": (_int, [ctypes.POINTER(_int), ctypes.POINTER(_int)]),
    "snarkv_last_error": (_cp, []),
    "snarkv_version": (_cp, []),
    "snarkv_g1_msm_naive": (_int, [_vp, _cp, _cp, _sz, _u32, _vp]),
    "snarkv_g1_msm_batched": (_int, [_vp, _cp, _cp, _vp, _sz, _u32, _vp]),
    "snarkv_g1_msm_pippenger": (_int, [_vp, _cp, _cp, _sz, _u32, _vp]),
    "snarkv_g1_msm_pippenger_dev": (_int, [_vp, _vp, _vp, _sz, _int, _vp]),
    "snarkv_g1_msm_pippenger_many_dev": (_int, [_vp, _sz, ctypes.POINTER(_vp), ctypes.POINTER(_vp), ctypes.POINTER(_sz), _int, _vp]),
    "snarkv_g1_msm_pippenger_many_partial_dev": (_int, [_vp, _sz, ctypes.POINTER(_vp), ctypes.POINTER(_vp), ctypes.POINTER(_sz), _int, _vp]),
    "snarkv_g1_msm_batched_dev": (_int, [_vp, _vp, _vp, _vp, _sz, _sz, _vp]),
    "snarkv_g1_msm_pippenger_partial_dev": (_int, [_vp, _vp, _vp, _sz, _int, _vp]),
    "snarkv_g1_fold_partials_dev": (_int, [_vp, _vp, _sz, _vp]),
    "snarkv_g1_fold_partials_many_dev": (_int, [_vp, _vp, _sz, _sz, _vp]),
    "snarkv_dk_create": (_int, [_vp, _cp, _cp, _cp, _u32, _pp]),
    "snarkv_dk_destroy": (None, [_vp]),
    "snarkv_kzg_decide": (_int, [_vp, _vp, _cp, _u32]),
    "snarkv_kzg_decide_batch": (_int, [_vp, _vp, _cp, _sz, _u32, _vp]),
    "snarkv_kzg_decide_batch_dev": (_int, [_vp, _vp, _vp, _sz, _vp]),
    "snarkv_kzg_pairing_value": (_int, [_vp, _vp, _cp, _vp]),
    "bn254_g1_msm_naive": (_int, [_cp, _cp, _sz, _vp]),
    "bn254_g1_msm_batched": (_int, [_cp, _cp, _vp, _sz, _vp]),
    "bn254_g1_msm_pippenger": (_int, [_cp, _cp, _sz, _vp]),
    "bn254_kzg_decide": (_int, [_cp, _cp, _cp, _cp]),
    "bn254_kzg_decide_batch": (_int, [_cp, _cp, _cp, _cp, _sz, _vp]),
    "snarkv_g1_validate": (_int, [_vp, _cp, _sz]),
    "bn254_g1_validate": (_int, [_cp, _sz]),
    "bn254_kzg_dk_create": (_int, [_cp, _cp, _cp, _pp]),
    "bn254_kzg_dk_decide_batch": (_int, [_vp, _cp, _sz, _vp]),
    "snarkv_sample_scalars_dev": (_int, [_vp, ctypes.c_uint64, ctypes.c_uint64, _sz, _vp]),
    "snarkv_sample_points_dev": (_int, [_vp, ctypes.c_uint64, ctypes.c_uint64, _sz, _vp]),
    "snarkv_ubench_valu": (_int, [_vp, _int, _int, ctypes.POINTER(ctypes.c_double)]),
    "snarkv_ctx_set_throughput_hint": (_int, [_vp, _int]),
    "snarkv_ctx_set_flags": (_int, [_vp, _u32]),
    "snarkv_g1_msm_pippenger_many": (_int, [_vp, _sz, _vp, _vp, _vp, _u32, _vp]),
    "snarkv_host_register": (_int, [_vp, _sz]),
    "snarkv_host_unregister": (_int, [_vp]),
    "snarkv_ctx_get_flags": (_u32, [_vp]),
    "bn254_set_flags": (_int, [_u32]),
    "snarkv_g1_msm_launch_points": (_int, [_sz, ctypes.POINTER(_sz)]),
    "snarkv_g1_msm_launch_points_ex": (_int, [_sz, _int, ctypes.POINTER(_sz)]),
    "snarkv_mgpu_create": (_int, [ctypes.POINTER(_int), _int, _pp]),
    "snarkv_mgpu_destroy": (None, [_vp]),
    "snarkv_mgpu_size": (_int, [_vp]),
    "snarkv_mgpu_ctx": (_vp, [_vp, _int]),
    "snarkv_mgpu_shard": (_int, [_vp, _sz, _int, ctypes.POINTER(_sz), ctypes.POINTER(_sz)]),
    "snarkv_mgpu_peer_access": (_int, [_vp, ctypes.POINTER(_int), ctypes.POINTER(_int), ctypes.POINTER(_int)]),
    "snarkv_mgpu_set_transport": (_int, [_vp, _int]),
    "snarkv_mgpu_result_dev": (_vp, [_vp, _int]),
    "snarkv_g1_msm_pippenger_mgpu": (_int, [_vp, _cp, _cp, _sz, _int, _vp]),
    "snarkv_g1_msm_pippenger_mgpu_dev": (_int, [_vp, ctypes.POINTER(_vp), ctypes.POINTER(_vp), ctypes.POINTER(_sz), _int, _int, _vp]),
    "snarkv_g1_msm_pippenger_many_mgpu_dev": (_int, [_vp, _sz, ctypes.POINTER(_vp), ctypes.POINTER(_vp), ctypes.POINTER(_sz), _int, _vp]),
    "snarkv_mgpu_results_many_dev": (_vp, [_vp, _int]),
    "snarkv_kzg_decide_batch_mgpu": (_int, [_vp, _cp, _cp, _cp, _cp, _sz, _vp]),
    "snarkv_g1_msm_bucket_geometry": (_int, [_sz, _int, ctypes.POINTER(_u32), ctypes.POINTER(_u32), ctypes.POINTER(_u32)]),
    "snarkv_g1_msm_fill_buckets_dev": (_int, [_vp, _vp, _vp, _sz, _int, _vp]),
    "snarkv_g1_buckets_add_dev": (_int, [_vp, _vp, _vp, _sz]),
    "snarkv_g1_buckets_reduce_dev": (_int, [_vp, _vp, _u32, _u32, _u32, _vp]),
    "snarkv_ipa_dk_create": (_int, [_vp, _vp, _sz, ctypes.POINTER(_vp)]),
    "snarkv_ipa_dk_create_shard": (_int, [_vp, _vp, _sz, _u32, _sz, ctypes.POINTER(_vp)]),
    "snarkv_ipa_commit_partial_dev": (_int, [_vp, _vp, _vp, _vp]),
    "snarkv_ipa_dk_destroy": (None, [_vp]),
    "snarkv_ipa_dk_k": (_u32, [_vp]),
    "snarkv_ipa_decide_batch": (_int, [_vp, _vp, _vp, _vp, _sz, _vp]),
    "bn254_ipa_dk_create": (_int, [_vp, _sz, ctypes.POINTER(_vp)]),
    "bn254_ipa_decide_batch": (_int, [_vp, _vp, _vp, _sz, _vp]),
    "snarkv_poseidon_create": (_int, [_vp, _u32, _u32, _u32, _u32, _cp, _cp, _cp, _cp, _cp, _cp, _cp, _pp]),
    "snarkv_poseidon_destroy": (None, [_vp]),
    "snarkv_poseidon_transcript_batch": (_int, [_vp, _vp, _cp, _sz, _sz, _vp, _sz, _vp]),
    "snarkv_poseidon_transcript_batch_dev": (_int, [_vp, _vp, _vp, _sz, _sz, _vp, _sz, _vp]),
    "bn254_poseidon_create": (_int, [_u32, _u32, _u32, _u32, _cp, _cp, _cp, _cp, _cp, _cp, _cp, _pp]),
    "bn254_poseidon_transcript_batch": (_int, [_vp, _cp, _sz, _sz, _vp, _sz, _vp]),
    "snarkv_poseidon_read_batch": (_int, [_vp, _vp, _cp, _sz, _sz, _cp, _sz, _vp, _sz, _vp, _sz, _vp, _sz, _vp, _vp, _vp]),
    "bn254_poseidon_read_batch": (_int, [_vp, _cp, _sz, _sz, _cp, _sz, _vp, _sz, _vp, _sz, _vp, _sz, _vp, _vp, _vp]),
    "snarkv_set_stage_timing": (_int, [_vp, _int]),
    "snarkv_get_stage_timing": (_int, [_vp, _vp]),
}

_lib = None


def lib_path():
    # SNARKV_AMD_LIB: alternate build of the same library (A/B runs of kernel variants)
    return os.environ.get("SNARKV_AMD_LIB") or os.path.join(HERE, _LIB_NAME)


def load_library():
    """Loads libsnarkv_amd.so; raises (loudly) when it has not been built."""
    global _lib
    if _lib is not None:
        return _lib
    path = lib_path()
    if not os.path.exists(path):
        raise ImportError(
            "%s is missing: build it with `python snark-verifier_amd/build.py` "
            "(hipcc --offload-arch=gfx950).  There is no CPU fallback." % path
        )
    # One HIP runtime per process: PyTorch-ROCm bundles its own libamdhip64.so.7
    # (same SONAME as /opt/rocm's).  If ours were loaded first, torch would bind
    # to it and fail to see the GPU; loading torch first makes this library bind
    # to the runtime torch already brought in, so device pointers and streams
    # can be shared between the two.
    try:
        import torch  # noqa: F401
    except Exception:  # torch is plumbing, not a requirement of the C ABI
        pass
    lib = ctypes.CDLL(path)
    for name, (res, args) in _SIGNATURES.items():
        fn = getattr(lib, name)  # AttributeError if the header and the library drift
        fn.restype = res
        fn.argtypes = args
    _lib = lib
    return lib


def last_error():
    """text of the calling thread's last library error (`snarkv_last_error()`)"""
    return load_library().snarkv_last_error().decode(errors="replace")


def _check(rc):
    if rc < 0:
        raise SnarkvError(rc, load_library().snarkv_last_error().decode(errors="replace"))
    return rc


def _as_bytes(x):
    if isinstance(x, ctypes.Array):  # Context.host_buffer: pinned memory, passed through by address
        return x
    if isinstance(x, (bytes, bytearray, memoryview)):
        return bytes(x)
    tb = getattr(x, "tobytes", None)  # numpy
    if tb is not None:
        return tb()
    raise TypeError("expected bytes-like, got %r" % type(x))


class MultiGpu:
    """Single-process multi-GPU handle (include/snarkv_amd.h "multi-GPU in ONE process"): one rank per entry of
    `devices` (a device may repeat).  `msm_pippenger(scalars, points, variant)` and `decide_batch(...)` shard, run
    and combine inside the library."""

    POINT_SHARDED, BUCKET_SHARDED = 0, 1

    def __init__(self, devices):
        self._lib = load_library()
        arr = (ctypes.c_int * len(devices))(*devices)
        self._h = ctypes.c_void_p()
        _check(self._lib.snarkv_mgpu_create(arr, len(devices), ctypes.byref(self._h)))
        self.world = len(devices)

    def close(self):
        if self._h:
            self._lib.snarkv_mgpu_destroy(self._h)
            self._h = None

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass

    PEER_COPY, RCCL = 0, 1

    def peer_access(self):
        """(enabled, unavailable, failed) directed pairs of distinct devices: direct xGMI access on / not offered / refused"""
        a, b, c = ctypes.c_int(0), ctypes.c_int(0), ctypes.c_int(0)
        _check(self._lib.snarkv_mgpu_peer_access(self._h, ctypes.byref(a), ctypes.byref(b), ctypes.byref(c)))
        return a.value, b.value, c.value

    def set_transport(self, transport):
        """PEER_COPY (default) or RCCL (`ncclCommInitAll` + one grouped `ncclAllGather` of the 144-byte partials)"""
        _check(self._lib.snarkv_mgpu_set_transport(self._h, transport))

    def result_dev(self, rank):
        """device pointer of rank's copy of the last MSM's 64-byte result (every rank holds it: all-reduce semantics)"""
        return self._lib.snarkv_mgpu_result_dev(self._h, rank)

    def shard(self, n_total, rank):
        lo, hi = ctypes.c_size_t(0), ctypes.c_size_t(0)
        _check(self._lib.snarkv_mgpu_shard(self._h, n_total, rank, ctypes.byref(lo), ctypes.byref(hi)))
        return lo.value, hi.value

    def rank_context(self, rank):
        """Borrowed `Context` of a rank (owned by the handle): to sample / stage that rank's shard on its device."""
        c = Context.__new__(Context)
        c._lib = self._lib
        c._h = ctypes.c_void_p(self._lib.snarkv_mgpu_ctx(self._h, rank))
        c._borrowed = True
        c.ordered = False  # (a rank of the handle, possibly on another device than torch's current one: the caller orders)
        return c

    def msm_pippenger(self, scalars, points, variant=0):
        scalars, points = _as_bytes(scalars), _as_bytes(points)
        n = len(scalars) // 32
        if len(points) != 64 * n:
            raise SnarkvError(SNARKV_ERR_LENGTH, "scalars / points length mismatch (reference: assert_eq!, msm.rs:309)")
        out = ctypes.create_string_buffer(64)
        _check(self._lib.snarkv_g1_msm_pippenger_mgpu(self._h, scalars, points, n, variant, out))
        return out.raw

    def msm_pippenger_dev(self, d_scalars, d_points, counts, window_bits=0, variant=0):
        w = self.world
        ds = (ctypes.c_void_p * w)(*[int(x) if x else None for x in d_scalars])
        dp = (ctypes.c_void_p * w)(*[int(x) if x else None for x in d_points])
        cn = (ctypes.c_size_t * w)(*counts)
        out = ctypes.create_string_buffer(64)
        _check(self._lib.snarkv_g1_msm_pippenger_mgpu_dev(self._h, ds, dp, cn, window_bits, variant, out))
        return out.raw

    def msm_pippenger_many_dev(self, d_scalars, d_points, counts, window_bits=0):
        """A batch of K MSMs, every one sharded over the ranks, ONE exchange for the batch.  `d_scalars[g][j]` /
        `d_points[g][j]` = rank g's shard of job j (device pointers on rank g's device), `counts[g][j]` its points.
        Returns the K affine results (64 bytes each)."""
        w, k = self.world, len(counts[0])
        flat = lambda rows: [rows[g][j] for g in range(w) for j in range(k)]  # noqa: E731
        ds = (ctypes.c_void_p * (w * k))(*[int(x) if x else None for x in flat(d_scalars)])
        dp = (ctypes.c_void_p * (w * k))(*[int(x) if x else None for x in flat(d_points)])
        cn = (ctypes.c_size_t * (w * k))(*flat(counts))
        out = ctypes.create_string_buffer(64 * k)
        _check(self._lib.snarkv_g1_msm_pippenger_many_mgpu_dev(self._h, k, ds, dp, cn, window_bits, out))
        return [out.raw[64 * j:64 * j + 64] for j in range(k)]

    def results_many_dev(self, rank):
        """device pointer of rank's copy of the last batch's results (K x 64 bytes)"""
        return self._lib.snarkv_mgpu_results_many_dev(self._h, rank)

    def decide_batch(self, g1, g2, s_g2, accs):
        accs = _as_bytes(accs)
        m = len(accs) // 128
        ok = ctypes.create_string_buffer(max(m, 1))
        allok = _check(self._lib.snarkv_kzg_decide_batch_mgpu(self._h, _as_bytes(g1), _as_bytes(g2), _as_bytes(s_g2), accs, m, ok))
        return bool(allok), [b != 0 for b in ok.raw[:m]]


class DecidingKey:
    """`KzgDecidingKey { svk.g, g2, s_g2 }` (reference
    snark-verifier/src/pcs/kzg/decider.rs:6-42) with its G2 line tables
    resident in HBM."""

    def __init__(self, ctx, g1, g2, s_g2, flags=0):
        self._lib = load_library()
        self._h = ctypes.c_void_p()
        g1, g2, s_g2 = _as_bytes(g1), _as_bytes(g2), _as_bytes(s_g2)
        assert len(g1) == 64 and len(g2) == 128 and len(s_g2) == 128
        _check(self._lib.snarkv_dk_create(ctx._h, g1, g2, s_g2, flags, ctypes.byref(self._h)))

    def close(self):
        if self._h:
            self._lib.snarkv_dk_destroy(self._h)
            self._h = ctypes.c_void_p()

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass


class IpaDecidingKey:
    """Device-resident committing key of `IpaDecidingKey` (reference pcs/ipa/decider.rs:5-9;
    include/snarkv_amd.h `snarkv_ipa_dk_create`): `g` = 2^k points, 64 bytes each."""

    def __init__(self, ctx, g, k=None, first=0):
        """All 2^k points, or -- with `k` and `first` -- the shard [first, first + len(g)/64) of a 2^k-point key."""
        self._lib = load_library()
        self._h = ctypes.c_void_p()
        g = _as_bytes(g)
        assert len(g) % 64 == 0
        if k is None:
            _check(self._lib.snarkv_ipa_dk_create(ctx._h, g if g else b"\x00", len(g) // 64, ctypes.byref(self._h)))
        else:
            _check(self._lib.snarkv_ipa_dk_create_shard(ctx._h, g if g else b"\x00", len(g) // 64, k, first,
                                                        ctypes.byref(self._h)))
        self.k = self._lib.snarkv_ipa_dk_k(self._h)

    def close(self):
        if self._h:
            self._lib.snarkv_ipa_dk_destroy(self._h)
            self._h = ctypes.c_void_p()

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass


class PoseidonSpec:
    """Device-resident tables of one Poseidon instance (include/snarkv_amd.h
    `snarkv_poseidon_create`); `tables` = dict of the optimised schedule, every
    entry a list (of lists) of integers < r:
    start [(r_f/2+1) x t], partial [r_p], end [(r_f/2-1) x t], mds [t x t],
    pre_sparse_mds [t x t], sparse_rows [r_p x t], sparse_col_hats [r_p x (t-1)]."""

    def __init__(self, ctx, t, rate, r_f, r_p, tables):
        self._lib = load_library()
        self._h = ctypes.c_void_p()
        self.t, self.rate, self.r_f, self.r_p = t, rate, r_f, r_p

        def flat(x):
            out = b""
            for row in x:
                if isinstance(row, (list, tuple)):
                    out += b"".join(int(v).to_bytes(32, "little") for v in row)
                else:
                    out += int(row).to_bytes(32, "little")
            return out if out else b"\x00"

        _check(self._lib.snarkv_poseidon_create(
            ctx._h, t, rate, r_f, r_p, flat(tables["start"]), flat(tables["partial"]), flat(tables["end"]),
            flat(tables["mds"]), flat(tables["pre_sparse_mds"]), flat(tables["sparse_rows"]),
            flat(tables["sparse_col_hats"]), ctypes.byref(self._h)))

    def close(self):
        if self._h:
            self._lib.snarkv_poseidon_destroy(self._h)
            self._h = ctypes.c_void_p()

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass


def stream_handle(stream, device=None):
    """hipStream_t as an int: a raw handle stays, a `torch.cuda.Stream` gives its `cuda_stream`, None = torch's CURRENT
    stream ON `device` (the context's: the ordering calls want a stream of the context's own device; 0 = the legacy
    default stream, which is what torch runs on unless told otherwise)."""
    if stream is None:
        import torch

        return int(torch.cuda.current_stream(device).cuda_stream)
    if isinstance(stream, int):
        return stream
    return int(stream.cuda_stream)


class Context:
    """One HIP stream + scratch (see include/snarkv_amd.h "Threading").

    `stream`: a HIP stream handle (e.g. `torch.cuda.Stream().cuda_stream`) the context
    launches on; None (or the NULL handle of torch's legacy default stream) makes the
    context create a private non-blocking stream.  `_dev` calls are asynchronous on
    that stream; against torch's stream they are ordered by events (`wait_stream` /
    `stream_wait`, i.e. `snarkv_ctx_wait_stream` / `snarkv_stream_wait_ctx`) -- automatically
    when `ordered` (the default for a private stream), by the caller otherwise (or run inside
    `torch.cuda.stream(s)` with the context created on `s.cuda_stream`)."""

    def __init__(self, device=0, stream=None, ordered=None):
        self._lib = load_library()
        self._h = ctypes.c_void_p()
        _check(self._lib.snarkv_ctx_create(int(device), ctypes.c_void_p(stream or 0), ctypes.byref(self._h)))
        self.device = int(device)
        # `ordered`: bracket every `*_dev` method with wait_stream() / stream_wait() against torch's CURRENT stream, so
        # that device tensors torch has just allocated / filled are ready when the context reads them and the outputs
        # are ready when torch touches them -- no host synchronisation.  Default: on for a context on its PRIVATE
        # stream (the caller did not think about streams: safe by default), off when the caller passed a stream (it
        # manages the ordering, e.g. runs inside `torch.cuda.stream(s)`: bench.py).  Pass ordered=False to keep SEVERAL
        # private-stream contexts concurrent from one host thread (ordering through torch's stream would chain them).
        self.ordered = (not stream) if ordered is None else bool(ordered)

    def close(self):
        if self._h and not getattr(self, "_borrowed", False):
            self._lib.snarkv_ctx_destroy(self._h)
        self._h = ctypes.c_void_p()

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass

    def sync(self):
        _check(self._lib.snarkv_ctx_sync(self._h))

    def wait_stream(self, stream=None):
        """`snarkv_ctx_wait_stream`: whatever this context enqueues next runs after everything queued on `stream` so far
        (a HIP stream handle, a `torch.cuda.Stream`, or None = torch's current stream).  No host synchronisation.  Call it
        after filling inputs with torch and before the `_dev` entry point that reads them."""
        _check(self._lib.snarkv_ctx_wait_stream(self._h, ctypes.c_void_p(stream_handle(stream, getattr(self, "device", None)))))

    def stream_wait(self, stream=None):
        """`snarkv_stream_wait_ctx`: whatever `stream` runs next (torch ops, an RCCL collective) runs after everything this
        context has enqueued so far.  No host synchronisation.  Call it after the `_dev` entry point whose output the
        stream reads."""
        _check(self._lib.snarkv_stream_wait_ctx(self._h, ctypes.c_void_p(stream_handle(stream, getattr(self, "device", None)))))

    @property
    def stream(self):
        """the hipStream_t (as an int) this context enqueues on (`snarkv_ctx_stream`)"""
        return int(self._lib.snarkv_ctx_stream(self._h) or 0)

    def g1_decompress(self, compressed):
        """Batch `G1Affine::from_bytes` (halo2.rs:260-273): n x 32 bytes -> (n x 64 bytes, [valid])."""
        c = _as_bytes(compressed)
        if len(c) % 32:
            raise SnarkvError(SNARKV_ERR_LENGTH, "compressed points are 32 bytes each")
        n = len(c) // 32
        out = ctypes.create_string_buffer(max(64 * n, 1))
        ok = ctypes.create_string_buffer(max(n, 1))
        _check(self._lib.snarkv_g1_decompress(self._h, c if n else b"\x00", n, out, ok))
        return out.raw[: 64 * n], [b != 0 for b in ok.raw[:n]]

    def host_buffer(self, slot, nbytes):
        """Pinned host memory owned by the context (`snarkv_ctx_host_buffer`): a ctypes char array over it
        (accepted by the msm_* methods in place of bytes).  Inputs
        assembled there reach the device by DMA when passed to the host-pointer entry points."""
        p = ctypes.c_void_p()
        _check(self._lib.snarkv_ctx_host_buffer(self._h, int(slot), int(nbytes), ctypes.byref(p)))
        return (ctypes.c_char * int(nbytes)).from_address(p.value)

    def msm_pippenger_many_host(self, scalars, points, counts, flags=0):
        """`snarkv_g1_msm_pippenger_many`: host-resident batch, uploads overlapped with the kernels.  scalars / points:
        lists of host addresses (ints, e.g. into `host_buffer`s) or bytes-like objects; returns the 64-byte results."""
        k = len(counts)
        keep = []

        def addr(x):
            if isinstance(x, int):
                return x
            b = (ctypes.c_char * len(x)).from_buffer_copy(x) if isinstance(x, (bytes, bytearray, memoryview)) else x
            keep.append(b)
            return ctypes.addressof(b)

        ps = (ctypes.c_void_p * k)(*[addr(x) for x in scalars])
        pp = (ctypes.c_void_p * k)(*[addr(x) for x in points])
        cn = (ctypes.c_size_t * k)(*[int(c) for c in counts])
        out = ctypes.create_string_buffer(max(64 * k, 1))
        _check(self._lib.snarkv_g1_msm_pippenger_many(self._h, k, ps, pp, cn, int(flags), out))
        return [out.raw[64 * i:64 * i + 64] for i in range(k)]

    def set_flags(self, flags):
        """Default flags of the context (`snarkv_ctx_set_flags`): SNARKV_FLAG_MONTGOMERY makes every call on it -- the
        device-resident ones and the samplers included -- speak halo2curves' in-memory form."""
        _check(self._lib.snarkv_ctx_set_flags(self._h, int(flags)))

    def get_flags(self):
        return int(self._lib.snarkv_ctx_get_flags(self._h))

    def set_throughput_hint(self, enabled=True):
        """Several MSMs in flight on several contexts: longer runs per lane (less work per MSM, longer single-MSM latency)."""
        _check(self._lib.snarkv_ctx_set_throughput_hint(self._h, 1 if enabled else 0))

    # ---- host-buffer entry points ------------------------------------
    def msm_naive(self, scalars, points, flags=0):
        """`NativeLoader::multi_scalar_multiplication` (native.rs:61-71)."""
        s, p = _as_bytes(scalars), _as_bytes(points)
        if len(s) % 32 or len(p) % 64 or len(s) // 32 != len(p) // 64:
            raise SnarkvError(SNARKV_ERR_LENGTH, "scalars/points length mismatch (reference: assert_eq!, msm.rs:309)")
        out = ctypes.create_string_buffer(64)
        _check(self._lib.snarkv_g1_msm_naive(self._h, s, p, len(s) // 32, flags, out))
        return out.raw

    def msm_batched(self, scalars, points, offsets, flags=0):
        import array

        s, p = _as_bytes(scalars), _as_bytes(points)
        offs = array.array("I", [int(o) for o in offsets])
        n_msm = len(offs) - 1
        if n_msm < 0 or len(s) % 32 or len(p) % 64 or len(s) // 32 != len(p) // 64:
            raise SnarkvError(SNARKV_ERR_LENGTH, "bad sizes")
        if n_msm >= 0 and len(offs) and offs[-1] != len(s) // 32:
            raise SnarkvError(SNARKV_ERR_LENGTH, "offsets[-1] != number of terms")
        out = ctypes.create_string_buffer(max(64 * n_msm, 1))
        addr, _ = offs.buffer_info()
        _check(self._lib.snarkv_g1_msm_batched(self._h, s, p, ctypes.c_void_p(addr), n_msm, flags, out))
        return out.raw[: 64 * n_msm]

    def msm_pippenger(self, scalars, points, flags=0):
        """`util::msm::multi_scalar_multiplication` (msm.rs:308-343), affine."""
        s, p = _as_bytes(scalars), _as_bytes(points)
        if len(s) % 32 or len(p) % 64 or len(s) // 32 != len(p) // 64:
            raise SnarkvError(SNARKV_ERR_LENGTH, "scalars/points length mismatch (reference: assert_eq!, msm.rs:309)")
        out = ctypes.create_string_buffer(64)
        _check(self._lib.snarkv_g1_msm_pippenger(self._h, s, p, len(s) // 32, flags, out))
        return out.raw

    def decide(self, dk, acc, flags=0):
        acc = _as_bytes(acc)
        assert len(acc) == 128
        return bool(_check(self._lib.snarkv_kzg_decide(self._h, dk._h, acc, flags)))

    def decide_batch(self, dk, accs, flags=0):
        accs = _as_bytes(accs)
        assert len(accs) % 128 == 0
        m = len(accs) // 128
        ok = ctypes.create_string_buffer(max(m, 1))
        allok = _check(self._lib.snarkv_kzg_decide_batch(self._h, dk._h, accs, m, flags, ok))
        return bool(allok), [b != 0 for b in ok.raw[:m]]

    def pairing_value(self, dk, acc):
        acc = _as_bytes(acc)
        out = ctypes.create_string_buffer(384)
        _check(self._lib.snarkv_kzg_pairing_value(self._h, dk._h, acc, out))
        return out.raw

    # ---- device-pointer entry points (ints from tensor.data_ptr()) ----
    def msm_pippenger_dev(self, d_scalars, d_points, n, d_out, window_bits=0):
        _check(self._lib.snarkv_g1_msm_pippenger_dev(self._h, d_scalars, d_points, n, window_bits, d_out))

    def msm_pippenger_many_dev(self, d_scalars, d_points, counts, d_out, window_bits=0):
        """`len(counts)` independent MSMs in one phase-ordered call: d_out[64 i ..] = MSM i (device pointers as ints)."""
        k = len(counts)
        ds = (ctypes.c_void_p * k)(*[int(x) for x in d_scalars])
        dp = (ctypes.c_void_p * k)(*[int(x) for x in d_points])
        cn = (ctypes.c_size_t * k)(*counts)
        _check(self._lib.snarkv_g1_msm_pippenger_many_dev(self._h, k, ds, dp, cn, window_bits, d_out))

    def msm_pippenger_many_partial_dev(self, d_scalars, d_points, counts, d_partials, window_bits=0):
        """the same with projective partials out (G1_PARTIAL_BYTES each): this rank's shard of `len(counts)` multi-GPU MSMs"""
        k = len(counts)
        ds = (ctypes.c_void_p * k)(*[int(x) for x in d_scalars])
        dp = (ctypes.c_void_p * k)(*[int(x) for x in d_points])
        cn = (ctypes.c_size_t * k)(*counts)
        _check(self._lib.snarkv_g1_msm_pippenger_many_partial_dev(self._h, k, ds, dp, cn, window_bits, d_partials))

    def msm_pippenger_partial_dev(self, d_scalars, d_points, n, d_partial, window_bits=0):
        _check(self._lib.snarkv_g1_msm_pippenger_partial_dev(self._h, d_scalars, d_points, n, window_bits, d_partial))

    def fold_partials_dev(self, d_partials, count, d_out):
        _check(self._lib.snarkv_g1_fold_partials_dev(self._h, d_partials, count, d_out))

    def fold_partials_many_dev(self, d_partials, count, jobs, d_out):
        """`jobs` folds of `count` partials each ([job][count] layout) in one launch"""
        _check(self._lib.snarkv_g1_fold_partials_many_dev(self._h, d_partials, count, jobs, d_out))

    def msm_batched_dev(self, d_scalars, d_points, d_offsets, n_msm, n_terms, d_out):
        _check(self._lib.snarkv_g1_msm_batched_dev(self._h, d_scalars, d_points, d_offsets, n_msm, n_terms, d_out))

    def decide_batch_dev(self, dk, d_accs, m, d_ok):
        _check(self._lib.snarkv_kzg_decide_batch_dev(self._h, dk._h, d_accs, m, d_ok))

    def sample_scalars_dev(self, seed, n, d_out, first=0):
        _check(self._lib.snarkv_sample_scalars_dev(self._h, seed, first, n, d_out))

    def sample_points_dev(self, seed, n, d_out, first=0):
        _check(self._lib.snarkv_sample_points_dev(self._h, seed, first, n, d_out))

    @staticmethod
    def launch_points(n, window_bits=0):
        """points one launch of the Pippenger kernels processes for an n-point MSM (n, or the chunk of the chunk pipeline)"""
        v = ctypes.c_size_t(0)
        _check(load_library().snarkv_g1_msm_launch_points_ex(n, window_bits, ctypes.byref(v)))
        return v.value

    @staticmethod
    def bucket_geometry(n_total, window_bits=0):
        """(c, windows, buckets_per_window) every rank must use for a bucket-sharded MSM of n_total points."""
        c, w, b = _u32(0), _u32(0), _u32(0)
        _check(load_library().snarkv_g1_msm_bucket_geometry(n_total, window_bits, ctypes.byref(c), ctypes.byref(w),
                                                            ctypes.byref(b)))
        return c.value, w.value, b.value

    def fill_buckets_dev(self, d_scalars, d_points, n, c, d_buckets):
        _check(self._lib.snarkv_g1_msm_fill_buckets_dev(self._h, d_scalars, d_points, n, c, d_buckets))

    def buckets_add_dev(self, d_dst, d_src, count):
        _check(self._lib.snarkv_g1_buckets_add_dev(self._h, d_dst, d_src, count))

    def buckets_reduce_dev(self, d_buckets, c, w0, wcount, d_partial):
        _check(self._lib.snarkv_g1_buckets_reduce_dev(self._h, d_buckets, c, w0, wcount, d_partial))

    def ipa_decide_batch(self, dk, xi, u):
        """`IpaAs::decide_all` semantics per accumulator (pcs/ipa/decider.rs:47-66): `xi` = m*k scalars
        (32 bytes LE each), `u` = m points (64 bytes each) -> list of m booleans."""
        xi, u = _as_bytes(xi), _as_bytes(u)
        m = len(u) // 64
        assert len(u) == 64 * m and len(xi) == 32 * dk.k * m
        ok = ctypes.create_string_buffer(max(m, 1))
        _check(self._lib.snarkv_ipa_decide_batch(self._h, dk._h, xi if xi else b"\x00", u if u else b"\x00", m, ok))
        return [b != 0 for b in ok.raw[:m]]

    def ipa_commit_partial_dev(self, dk, xi, d_partial):
        """This shard's part of commit(G, h(xi)) as a projective partial at device address `d_partial`."""
        xi = _as_bytes(xi)
        assert len(xi) == 32 * dk.k
        _check(self._lib.snarkv_ipa_commit_partial_dev(self._h, dk._h, xi, d_partial))

    def poseidon_transcript_batch(self, spec, elems, n, seg_len):
        """n transcripts: `elems` = n*L canonical 32-byte Fr, absorbed in len(seg_len) segments with a
        squeeze after each; returns n*len(seg_len) challenges (32 bytes LE each)."""
        import array

        elems = _as_bytes(elems)
        L = sum(seg_len)
        assert len(elems) == 32 * n * L
        segs = array.array("I", seg_len)
        addr, _ = segs.buffer_info()
        out = ctypes.create_string_buffer(32 * n * len(seg_len))
        _check(self._lib.snarkv_poseidon_transcript_batch(self._h, spec._h, elems if elems else b"\x00", n, L,
                                                          ctypes.c_void_p(addr), len(seg_len), out))
        return out.raw

    def poseidon_read_batch(self, spec, proofs, n, stride, lead, n_lead, layout, point_offsets, seg_len):
        """n proofs (stride bytes each) hashed where they are: layout[k] = kind << 28 | value names the source of absorbed
        element k (0 lead element, 1 scalar at that byte of the proof, 2 / 3 x / y of that point); returns (challenges,
        points64, ok) -- include/snarkv_amd.h snarkv_poseidon_read_batch."""
        import array

        proofs, lead = _as_bytes(proofs), _as_bytes(lead)
        assert len(proofs) == n * stride and len(lead) == 32 * n * n_lead and sum(seg_len) == len(layout)
        lay, offs, segs = array.array("I", layout), array.array("I", point_offsets), array.array("I", seg_len)
        P, S = len(point_offsets), len(seg_len)
        ch = ctypes.create_string_buffer(max(1, 32 * n * S))
        pts = ctypes.create_string_buffer(max(1, 64 * n * P))
        ok = ctypes.create_string_buffer(max(1, n * P))
        _check(self._lib.snarkv_poseidon_read_batch(
            self._h, spec._h, proofs, n, stride, lead if lead else b"\x00", n_lead, ctypes.c_void_p(lay.buffer_info()[0]), len(layout),
            ctypes.c_void_p(offs.buffer_info()[0]) if P else None, P, ctypes.c_void_p(segs.buffer_info()[0]), S, ch, pts, ok))
        return ch.raw[:32 * n * S], pts.raw[:64 * n * P], ok.raw[:n * P]

    def ubench_valu(self, which, iters=400):
        v = ctypes.c_double()
        _check(self._lib.snarkv_ubench_valu(self._h, which, iters, ctypes.byref(v)))
        return v.value

    def set_stage_timing(self, enabled=True):
        _check(self._lib.snarkv_set_stage_timing(self._h, 1 if enabled else 0))

    def get_stage_timing(self):
        arr = (ctypes.c_float * SNARKV_PIP_STAGES)()
        _check(self._lib.snarkv_get_stage_timing(self._h, arr))
        return dict(zip(PIP_STAGE_NAMES, [float(x) for x in arr]))


def _ordered_call(fn):
    import functools

    @functools.wraps(fn)
    def call(self, *a, **k):
        if not getattr(self, "ordered", False):
            return fn(self, *a, **k)
        self.wait_stream()
        try:
            return fn(self, *a, **k)
        finally:
            self.stream_wait()

    return call


for _name, _fn in list(vars(Context).items()):
    if _name.endswith("_dev") and callable(_fn):
        setattr(Context, _name, _ordered_call(_fn))
