"""Builds libsnarkv_amd.so (HIP kernels + C ABI) for gfx950, in-tree.

`hipcc --offload-arch=gfx950` cross-compiles without a GPU.  Objects land in
`snark-verifier_amd/build/`, the shared library next to this file so it travels
with the gpurun snapshot.  Rebuilds only what is stale.
"""
import os
import subprocess
import sys
from concurrent.futures import ThreadPoolExecutor

HERE = os.path.dirname(os.path.abspath(__file__))
CSRC = os.path.join(HERE, "csrc")
BUILD = os.path.join(HERE, "build")
LIB = os.path.join(HERE, "libsnarkv_amd.so")
UNITS = ["capi", "msm_naive", "msm_pippenger", "decider", "sample", "poseidon", "ipa", "mgpu", "decompress"]
HIPCC = os.environ.get("HIPCC", "/opt/rocm/bin/hipcc")
FLAGS = ["--offload-arch=gfx950", "-O3", "-std=c++17", "-fPIC", "-fno-gpu-rdc", "-Wno-unused-result"]
FLAGS += os.environ.get("SNARKV_EXTRA_FLAGS", "").split()


def _deps():
    hdrs = [os.path.join(CSRC, f) for f in os.listdir(CSRC) if f.endswith((".h", ".hpp", ".inc"))]
    hdrs.append(os.path.join(HERE, "..", "include", "snarkv_amd.h"))
    hdrs.append(os.path.join(HERE, "..", "include", "snarkv_pallas.h"))
    return max(os.path.getmtime(h) for h in hdrs)


def _compile(unit, verbose, extra=(), tag=""):
    src = os.path.join(CSRC, unit + ".hip")
    obj = os.path.join(BUILD, unit + tag + ".o")
    newest = max(os.path.getmtime(src), _deps())
    if os.path.exists(obj) and os.path.getmtime(obj) >= newest:
        return obj, False
    cmd = [HIPCC] + FLAGS + list(extra) + ["-c", src, "-o", obj]
    if verbose:
        cmd.append("-Rpass-analysis=kernel-resource-usage")
    r = subprocess.run(cmd, capture_output=True, text=True)
    with open(os.path.join(BUILD, unit + tag + ".log"), "w") as f:
        f.write(r.stdout + r.stderr)
    if r.returncode != 0:
        sys.stderr.write(r.stdout + r.stderr)
        raise RuntimeError("hipcc failed for %s" % unit)
    return obj, True


def build(verbose=False):
    os.makedirs(BUILD, exist_ok=True)
    # every unit of both libraries in one pool (hipcc is single-threaded per unit)
    jobs = [(u, (), "") for u in UNITS] + [(u, tuple(PALLAS_FLAGS), "_pallas") for u in PALLAS_UNITS]
    with ThreadPoolExecutor(max_workers=min(len(jobs), os.cpu_count() or 4)) as ex:
        res = list(ex.map(lambda j: _compile(j[0], verbose, j[1], j[2]), jobs))
    _link(LIB, res[:len(UNITS)], [])
    _link(PALLAS_LIB, res[len(UNITS):], ["-Wl,-Bsymbolic"])  # its internals never bind by symbol lookup across libraries
    build_host_driver()
    return LIB


def _link(lib, res, extra):
    if any(ch for _, ch in res) or not os.path.exists(lib):
        cmd = [HIPCC, "--offload-arch=gfx950", "-shared", "-fPIC"] + extra + ["-o", lib] + [o for o, _ in res]
        r = subprocess.run(cmd, capture_output=True, text=True)
        if r.returncode != 0:
            sys.stderr.write(r.stdout + r.stderr)
            raise RuntimeError("link failed: %s" % os.path.basename(lib))


# The pasta build of the curve-generic units (csrc/pallas.hip explains the flags).
PALLAS_UNITS = ["pallas", "msm_pippenger", "msm_naive", "ipa"]
PALLAS_FLAGS = ["-DSNARKV_CURVE_PALLAS", "-Dsnarkv=snarkv_pallas"]
PALLAS_LIB = os.path.join(HERE, "libsnarkv_pallas.so")


def build_pallas(verbose=False):
    os.makedirs(BUILD, exist_ok=True)
    with ThreadPoolExecutor(max_workers=len(PALLAS_UNITS)) as ex:
        res = list(ex.map(lambda u: _compile(u, verbose, PALLAS_FLAGS, "_pallas"), PALLAS_UNITS))
    _link(PALLAS_LIB, res, ["-Wl,-Bsymbolic"])
    return PALLAS_LIB


HOST = os.path.join(HERE, "host")
HOST_LIB = os.path.join(HERE, "libsnarkv_host.so")                 # product: C API of the host mirror (host/capi.cpp)
HOSTTEST_LIB = os.path.join(HERE, "libsnarkv_hosttest.so")         # test hooks only (host/test_driver.cpp)
HOST_PALLAS_LIB = os.path.join(HERE, "libsnarkv_hosttest_pallas.so")  # pasta flavour of the mirror, test hooks


def _host_stale(out, dev_lib):
    srcs = [os.path.join(HOST, f) for f in os.listdir(HOST)] + [os.path.join(os.path.dirname(HERE), "include", "snarkv_host.h")]
    newest = max([os.path.getmtime(f) for f in srcs] + [os.path.getmtime(dev_lib)])
    return not os.path.exists(out) or os.path.getmtime(out) < newest


def _gxx(out, src, extra, dev):
    # -mbmi2 -madx: mulx/adcx for the 4x64 Montgomery products (every x86-64 server CPU since 2015)
    cmd = ["g++", "-O3", "-mbmi2", "-madx", "-std=c++17", "-shared", "-fPIC"] + extra + ["-o", out, os.path.join(HOST, src),
           "-L" + HERE, "-l" + dev, "-pthread", "-Wl,-rpath,$ORIGIN"]
    r = subprocess.run(cmd, capture_output=True, text=True)
    if r.returncode != 0:
        sys.stderr.write(r.stdout + r.stderr)
        raise RuntimeError("host build failed: %s" % src)
    return out


def build_host_driver():
    """C++ host mirror (host/*.hpp): its C API (host/capi.cpp -> libsnarkv_host.so, include/snarkv_host.h), the test
    hooks (host/test_driver.cpp -> libsnarkv_hosttest.so) and the pasta flavour's test hooks, each linked against the
    device library next to it (rpath $ORIGIN).  Every target has its own staleness check."""
    jobs = []
    if _host_stale(HOST_LIB, LIB):
        jobs.append((HOST_LIB, "capi.cpp", [], "snarkv_amd"))
    if _host_stale(HOSTTEST_LIB, LIB):
        jobs.append((HOSTTEST_LIB, "test_driver.cpp", [], "snarkv_amd"))
    if os.path.exists(PALLAS_LIB) and _host_stale(HOST_PALLAS_LIB, PALLAS_LIB):
        # -Dsnarkv_host=...: its own C++ namespace -- the two flavours define the same inline functions and
        # `static constexpr` members with different constants, and C++17 inline variables are STB_GNU_UNIQUE
        # (bound process-wide even under RTLD_LOCAL) when both libraries sit in one process
        jobs.append((HOST_PALLAS_LIB, "test_driver_pallas.cpp",
                     ["-DSNARKV_HOST_PALLAS", "-Dsnarkv_host=snarkv_host_pallas", "-fno-gnu-unique"], "snarkv_pallas"))
    if jobs:
        with ThreadPoolExecutor(max_workers=len(jobs)) as ex:
            list(ex.map(lambda j: _gxx(*j), jobs))
    return HOST_LIB


def build_host_driver_pallas():
    build_host_driver()
    return HOST_PALLAS_LIB


if __name__ == "__main__":
    import time

    t = time.time()
    print(build(verbose="-v" in sys.argv), "built in %.1fs" % (time.time() - t))
