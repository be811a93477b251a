"""CPU: the C-ABI shared library loads and exports every symbol that
include/snarkv_amd.h declares; the ctypes table lists exactly those."""
import os
import re

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _header_functions():
    txt = open(os.path.join(ROOT, "include", "snarkv_amd.h")).read()
    txt = re.sub(r"/\*.*?\*/", "", txt, flags=re.S)
    names = re.findall(r"\b((?:snarkv|bn254)_[a-z0-9_]+)\s*\(", txt)
    return sorted(set(names))


def test_library_exports_every_declared_symbol():
    import snark_verifier_amd as sv
    from snark_verifier_amd import _lib

    lib = sv.load_library()
    declared = _header_functions()
    assert len(declared) >= 25
    for name in declared:
        assert hasattr(lib, name), name
    assert sorted(_lib._SIGNATURES) == declared


def test_no_cpu_fallback_without_device():
    import snark_verifier_amd as sv

    try:
        import torch

        has_gpu = torch.cuda.is_available()
    except Exception:
        has_gpu = False
    if has_gpu:
        pytest.skip("GPU present")
    with pytest.raises(sv.SnarkvError) as e:
        sv.Context(0)
    assert e.value.code == -4


def test_pallas_library_exports_every_declared_symbol():
    """The pasta build (include/snarkv_pallas.h -> libsnarkv_pallas.so) and, without a device, the same
    loud failure as the BN254 library."""
    import snark_verifier_amd as sv
    from snark_verifier_amd import pallas as PL

    txt = open(os.path.join(ROOT, "include", "snarkv_pallas.h")).read()
    txt = re.sub(r"/\*.*?\*/", "", txt, flags=re.S)
    declared = sorted(set(re.findall(r"\b((?:snarkv_)?pallas_[a-z0-9_]+)\s*\(", txt)))
    assert len(declared) == 22
    lib = PL.load_library()
    for name in declared:
        assert hasattr(lib, name), name
    assert b"pallas" in lib.snarkv_pallas_version()
    assert b"bn254" in sv.load_library().snarkv_version()
    try:
        import torch

        has_gpu = torch.cuda.is_available()
    except Exception:
        has_gpu = False
    if not has_gpu:
        with pytest.raises(sv.SnarkvError) as e:
            PL.PallasContext(0)
        assert e.value.code == -4
