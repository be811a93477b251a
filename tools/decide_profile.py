"""Dev tool: phase breakdown of k_decide.  Needs a library built with
SNARKV_EXTRA_FLAGS=-DSNARKV_DECIDE_PROFILE (stamps overwrite the Gt output)."""
import os, struct, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
import snark_verifier_amd as sv
ctx = sv.Context(0)
g2 = bytes.fromhex(
    "edf692d95cbdde46ddda5ef7d422436779445c5e66006a42761e1f12efde0018c212f3aeb785e49712e7a9353349aaf1255dfb31b7bf60723a480d9293938e19"
    "aa7dfa6601cce64c7bd3430c69e7d1e38f40cb8d8071ab4aeb6d8cdba55ec8125b9722d1dcdaac55f38eb37033314bbc95330c69ad999eec75f05f58d0890609")
g1 = (1).to_bytes(32, "little") + (2).to_bytes(32, "little")
dk = sv.DecidingKey(ctx, g1, g2, g2)
names = ["load+lines", "miller", "inversion", "easy part", "3 x exp_by_x", "hard part"]
for _ in range(3):
    raw = ctx.pairing_value(dk, g1 + g1)
st = struct.unpack("<7Q", raw[:56])
print("teams=%s" % os.environ.get("SNARKV_DECIDE_FORM", "auto"),
      {n: "%.1f us" % ((st[k + 1] - st[k]) / 100.0) for k, n in enumerate(names)}, "total %.1f us" % ((st[6] - st[0]) / 100.0))
