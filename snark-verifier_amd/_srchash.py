"""sha256 over the device sources (csrc/*.hip, *.h, *.inc, *.hpp, sorted by name): the identity of the kernels a
profile record was taken on.  tools/isa_stats.py and tools/pmc_traffic.py store it, bench.py recomputes it and only
quotes a record whose hash equals the tree's."""
import hashlib
import os

CSRC = os.path.join(os.path.dirname(os.path.abspath(__file__)), "csrc")


def kernel_source_hash():
    h = hashlib.sha256()
    for name in sorted(os.listdir(CSRC)):
        if name.endswith((".hip", ".h", ".inc", ".hpp")):
            h.update(name.encode() + b"\0")
            with open(os.path.join(CSRC, name), "rb") as f:
                h.update(f.read())
    return h.hexdigest()[:16]
