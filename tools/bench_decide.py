import sys, time, os
sys.path.insert(0, os.getcwd())
import torch, snark_verifier_amd as sv
ctx = sv.Context(0)
g2 = bytes.fromhex("edf692d95cbdde46ddda5ef7d422436779445c5e66006a42761e1f12efde0018c212f3aeb785e49712e7a9353349aaf1255dfb31b7bf60723a480d9293938e19aa7dfa6601cce64c7bd3430c69e7d1e38f40cb8d8071ab4aeb6d8cdba55ec8125b9722d1dcdaac55f38eb37033314bbc95330c69ad999eec75f05f58d0890609")
g1 = (1).to_bytes(32, "little") + (2).to_bytes(32, "little")
dk = sv.DecidingKey(ctx, g1, g2, g2)
one = torch.frombuffer(bytearray((g1 + g1) * 1024), dtype=torch.uint8).cuda()
oks = torch.zeros(1024, dtype=torch.uint8, device="cuda")
torch.cuda.synchronize()
for m in (1, 64, 1024):
    for _ in range(3): ctx.decide_batch_dev(dk, one.data_ptr(), m, oks.data_ptr()); ctx.sync()
    t0 = time.perf_counter()
    for _ in range(20): ctx.decide_batch_dev(dk, one.data_ptr(), m, oks.data_ptr()); ctx.sync()
    print("decide m=%d: %.3f ms" % (m, (time.perf_counter() - t0) / 20 * 1e3), end="; ")
print()
