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
    assert len(declared) == 25
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


def test_headers_are_plain_c_and_a_c_program_links(tmp_path):
    """The drop-in boundary is a C ABI: every header under include/ compiles as strict C99 (no C++ types, no torch types in
    a signature), and a C program that names one entry point of each family links against the libraries with gcc alone."""
    import subprocess

    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    inc, libdir = os.path.join(root, "include"), os.path.join(root, "snark-verifier_amd")
    for h in ("snarkv_amd.h", "snarkv_host.h", "snarkv_pallas.h"):
        r = subprocess.run(["gcc", "-std=c99", "-Wall", "-Wextra", "-pedantic", "-Werror", "-fsyntax-only", "-x", "c", os.path.join(inc, h)],
                           capture_output=True, text=True)
        assert r.returncode == 0, h + ": " + r.stderr
    src = tmp_path / "link.c"
    src.write_text('''#include "snarkv_amd.h"
#include "snarkv_host.h"
#include <stdio.h>
int main(void) {
  /* addresses only: no device call without a GPU */
  void* f[] = {(void*)snarkv_ctx_create, (void*)snarkv_g1_msm_pippenger_many_dev, (void*)bn254_g1_msm_batched,
               (void*)bn254_set_thread_flags, (void*)snarkv_kzg_decide_batch, (void*)snarkv_g1_msm_pippenger_many_mgpu_dev,
               (void*)snarkv_host_aggregate, (void*)snarkv_ctx_wait_stream, (void*)snarkv_stream_wait_ctx, (void*)snarkv_ctx_stream};
  /* argument checks that need no device: a NULL context is SNARKV_ERR_ARG, never a crash */
  if (snarkv_ctx_wait_stream(0, 0) != SNARKV_ERR_ARG || snarkv_stream_wait_ctx(0, 0) != SNARKV_ERR_ARG || snarkv_ctx_stream(0) != 0) return 3;
  printf("%s %d\\n", snarkv_version(), (int)(sizeof f / sizeof f[0]));
  return 0;
}
''')
    exe = tmp_path / "link"
    r = subprocess.run(["gcc", "-std=gnu99", "-I", inc, str(src), "-o", str(exe), "-L", libdir, "-lsnarkv_host", "-lsnarkv_amd",
                        "-Wl,-rpath," + libdir, "-Wl,-rpath-link," + libdir + ":/opt/rocm/lib"], capture_output=True, text=True)
    assert r.returncode == 0, r.stderr
    r = subprocess.run([str(exe)], capture_output=True, text=True, env=dict(os.environ, LD_LIBRARY_PATH="/opt/rocm/lib:" + os.environ.get("LD_LIBRARY_PATH", "")))
    assert r.returncode == 0 and r.stdout.split()[-1] == "10", r.stdout + r.stderr
