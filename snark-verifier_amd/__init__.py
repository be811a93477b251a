"""MI355X-native KZG accumulation hot path (after privacy-scaling-explorations/snark-verifier).

Thin Python binding over the C ABI in `include/snarkv_amd.h` (ctypes).  The
product is `libsnarkv_amd.so` -- hand-written HIP kernels for gfx950 -- and the
C++ host mirror in `host/`; Python only moves bytes and device pointers and
provides `torch.distributed` plumbing for the multi-GPU fold.

There is NO CPU fallback: importing works anywhere, but creating a `Context`
without the built library or without a HIP device raises.
"""
import os as _os

# HIP multiplexes a process's streams onto GPU_MAX_HW_QUEUES hardware queues (default 4) and kernels of streams that
# share a queue run one after the other.  The aggregation path is made of latency-bound launches (segmented small MSMs,
# one-workgroup pairing deciders) that only fill the GPU when many jobs overlap: with 16 queues 16 jobs in flight verify
# 2.9 x 10^5 proofs/s in 64-proof jobs against 1.15 x 10^5 with 4 (profiles/r03_agg_hw_queues.txt); the large-MSM batch
# (5 streams) is level to 1 % better.  Read by the HIP runtime at its first call, so it is set here, before the library
# is loaded; an explicit setting of the caller wins.  C / Rust callers set it themselves before their first HIP call
# (INTEGRATION.md): the library does NOT touch the environment (csrc/capi.hip).
_os.environ.setdefault("GPU_MAX_HW_QUEUES", "16")

from ._lib import (  # noqa: F401,E402
    Context,
    DecidingKey,
    IpaDecidingKey,
    MultiGpu,
    PoseidonSpec,
    SnarkvError,
    last_error,
    lib_path,
    load_library,
    SNARKV_FLAG_VALIDATE,
    SNARKV_FLAG_MONTGOMERY,
    SNARKV_HOST_BUFFERS,
    SNARKV_ERR_EMPTY,
    SNARKV_ERR_LENGTH,
    SNARKV_ERR_ENCODING,
    SNARKV_ERR_DEVICE,
    SNARKV_ERR_ARG,
    PIP_STAGE_NAMES,
    G1_PARTIAL_BYTES,
)

from . import host_api  # noqa: F401,E402  (C API of the C++ host mirror: include/snarkv_host.h)

__all__ = [
    "host_api",
    "Context",
    "DecidingKey",
    "IpaDecidingKey",
    "MultiGpu",
    "PoseidonSpec",
    "SnarkvError",
    "last_error",
    "lib_path",
    "load_library",
    "SNARKV_FLAG_VALIDATE",
    "SNARKV_FLAG_MONTGOMERY",
    "SNARKV_HOST_BUFFERS",
    "SNARKV_ERR_EMPTY",
    "SNARKV_ERR_LENGTH",
    "SNARKV_ERR_ENCODING",
    "SNARKV_ERR_DEVICE",
    "SNARKV_ERR_ARG",
    "PIP_STAGE_NAMES",
    "G1_PARTIAL_BYTES",
]
