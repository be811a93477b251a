"""CPU: libsnarkv_host.so (the C API of the C++ host mirror) loads, exports every symbol include/snarkv_host.h
declares, the ctypes table lists exactly those, and argument errors come back as codes (no compute without a GPU)."""
import ctypes
import os
import re

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def test_host_library_exports_every_declared_symbol():
    from snark_verifier_amd import host_api as H

    txt = open(os.path.join(ROOT, "include", "snarkv_host.h")).read()
    txt = re.sub(r"/\*.*?\*/", "", txt, flags=re.S)
    declared = sorted(set(re.findall(r"\b(snarkv_host_[a-z0-9_]+)\s*\(", txt)))
    assert len(declared) >= 18
    lib = H.load_library()
    for name in declared:
        assert hasattr(lib, name), name
    assert sorted(H._SIGNATURES) == declared
    # the product library carries no test hooks, the test-hook library no product API
    assert not hasattr(lib, "hd_aggregate_end_to_end")
    T = ctypes.CDLL(os.path.join(ROOT, "snark-verifier_amd", "libsnarkv_hosttest.so"))
    assert hasattr(T, "hd_msm_evaluate") and not hasattr(T, "snarkv_host_aggregate")


def test_host_api_argument_errors_without_device():
    import pytest

    from snark_verifier_amd import host_api as H

    lib = H.load_library()
    h = ctypes.c_void_p()
    assert lib.snarkv_host_protocol_parse(None, 0, 0, ctypes.byref(h)) == H.ERR_ARG
    assert lib.snarkv_host_protocol_parse(b"\x01\x02", 2, 0, ctypes.byref(h)) == H.ERR_PANIC  # truncated protocol bytes
    assert b"truncated" in lib.snarkv_host_last_error()
    assert lib.snarkv_host_protocol_parse(b"{}", 2, 7, ctypes.byref(h)) == H.ERR_ARG
    with pytest.raises(H.HostError) as e:
        H.Protocol(b'{"domain": 1}', H.PROTOCOL_SERDE_JSON)
    assert e.value.code == H.ERR_PANIC
    assert lib.snarkv_host_kzg_decide(None, None) == H.ERR_ARG
