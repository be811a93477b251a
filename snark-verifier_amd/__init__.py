"""MI355X-native KZG accumulation hot path (after privacy-scaling-explorations/snark-verifier).

Thin Python binding over the C ABI in `include/snarkv_amd.h` (ctypes).  The
product is `libsnarkv_amd.so` -- hand-written HIP kernels for gfx950 -- and the
C++ host mirror in `host/`; Python only moves bytes and device pointers and
provides `torch.distributed` plumbing for the multi-GPU fold.

There is NO CPU fallback: importing works anywhere, but creating a `Context`
without the built library or without a HIP device raises.
"""
from ._lib import (  # noqa: F401
    Context,
    DecidingKey,
    IpaDecidingKey,
    MultiGpu,
    PoseidonSpec,
    SnarkvError,
    lib_path,
    load_library,
    SNARKV_FLAG_VALIDATE,
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
    "lib_path",
    "load_library",
    "SNARKV_FLAG_VALIDATE",
    "SNARKV_HOST_BUFFERS",
    "SNARKV_ERR_EMPTY",
    "SNARKV_ERR_LENGTH",
    "SNARKV_ERR_ENCODING",
    "SNARKV_ERR_DEVICE",
    "SNARKV_ERR_ARG",
    "PIP_STAGE_NAMES",
    "G1_PARTIAL_BYTES",
]
