"""bn254 deciding key: time of snarkv_dk_create (k_validate_g2 + k_g2_prepare_w: both line tables) and of the context-free
bn254_kzg_decide on a cached key.  python tools/dk_create_time.py"""
import os
import sys
import time

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import snark_verifier_amd as sv

ctx = sv.Context(0)
g2 = bytes.fromhex("edf692d95cbdde46ddda5ef7d422436779445c5e66006a42761e1f12efde0018c212f3aeb785e49712e7a9353349aaf1255dfb31b7bf60723a480d9293938e19aa7dfa6601cce64c7bd3430c69e7d1e38f40cb8d8071ab4aeb6d8cdba55ec8125b9722d1dcdaac55f38eb37033314bbc95330c69ad999eec75f05f58d0890609")
g1 = (1).to_bytes(32, "little") + (2).to_bytes(32, "little")
for _ in range(3):
    sv.DecidingKey(ctx, g1, g2, g2).close()
t0 = time.perf_counter()
for _ in range(20):
    sv.DecidingKey(ctx, g1, g2, g2).close()
print("snarkv_dk_create (line tables of g2, -s_g2 on one wavefront): %.3f ms per key" % ((time.perf_counter() - t0) / 20 * 1e3))
lib = sv.load_library()
acc = g1 + g1
for _ in range(3):
    lib.bn254_kzg_decide(g1, g2, g2, acc)
t0 = time.perf_counter()
for _ in range(20):
    assert lib.bn254_kzg_decide(g1, g2, g2, acc) == 1
print("bn254_kzg_decide (context-free, same key: tables kept): %.3f ms per call" % ((time.perf_counter() - t0) / 20 * 1e3))
