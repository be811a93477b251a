"""dev tool (CPU only): microseconds per proof of the host half's phases (host/plonk.hpp), one thread, on the 64-proof
fixture -- what bench.py's `ms_fr_algebra_host` is made of.   python tools/host_phases.py [reps]"""
import ctypes
import os
import struct
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "tests"))
sys.path.insert(0, os.path.join(ROOT, "oracle"))
sys.path.insert(0, ROOT)
from hostfmt import load_host_lib

b = open(os.path.join(ROOT, "tests", "golden", "bench_plonk_gwc19_evm_64.bin"), "rb").read()
n, = struct.unpack_from("<I", b, 4)
off, parts = 8, []
for _ in range(3):
    ln, = struct.unpack_from("<I", b, off)
    parts.append(b[off + 4:off + 4 + ln])
    off += 4 + ln
dk = b[off:off + 320]
L = load_host_lib()
out = (ctypes.c_double * 7)()
reps = int(sys.argv[1]) if len(sys.argv) > 1 else 20
L.hd_plonk_host_phases.argtypes = [ctypes.c_char_p, ctypes.c_size_t, ctypes.c_char_p, ctypes.c_size_t, ctypes.c_char_p, ctypes.c_size_t,
                                   ctypes.c_uint32, ctypes.c_char_p, ctypes.c_int, ctypes.POINTER(ctypes.c_double)]
rc = L.hd_plonk_host_phases(parts[0], len(parts[0]), parts[1], len(parts[1]), parts[2], len(parts[2]), n, dk, reps, out)
names = ["read_proof", "CommonPolyEval", "evaluations_map", "commitments", "queries", "pcs_msms", "pairs"]
print("rc", rc, "us per proof:", {k: round(v, 2) for k, v in zip(names, out)}, "sum %.1f" % sum(out))
