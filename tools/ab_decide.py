"""A/B of the decide kernel forms (SNARKV_DECIDE_FORM = 1 one team / 2 two teams (round 3) / 3 program-driven), interleaved:
per batch size, `--rounds` passes over the forms, each pass the median-free mean of `--reps` device-only calls
(snarkv_kzg_decide_batch_dev + sync); prints per form the median over passes.  Dev tool (profiles/r04_ab_decide_wave.txt)."""
import argparse, os, statistics, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch, snark_verifier_amd as sv

ap = argparse.ArgumentParser()
ap.add_argument("--rounds", type=int, default=7)
ap.add_argument("--reps", type=int, default=20)
ap.add_argument("--sizes", default="1,16,64,256,1024")
ap.add_argument("--forms", default="3,1")
a = ap.parse_args()
ctx = sv.Context(0)
g2 = bytes.fromhex("edf692d95cbdde46ddda5ef7d422436779445c5e66006a42761e1f12efde0018c212f3aeb785e49712e7a9353349aaf1255dfb31b7bf60723a480d9293938e19aa7dfa6601cce64c7bd3430c69e7d1e38f40cb8d8071ab4aeb6d8cdba55ec8125b9722d1dcdaac55f38eb37033314bbc95330c69ad999eec75f05f58d0890609")
g1 = (1).to_bytes(32, "little") + (2).to_bytes(32, "little")
dk = sv.DecidingKey(ctx, g1, g2, g2)
one = torch.frombuffer(bytearray((g1 + g1) * 1024), dtype=torch.uint8).cuda()
oks = torch.zeros(1024, dtype=torch.uint8, device="cuda")
torch.cuda.synchronize()
forms = a.forms.split(",")
for m in [int(x) for x in a.sizes.split(",")]:
    res = {f: [] for f in forms}
    for _ in range(a.rounds):
        for f in forms:
            os.environ["SNARKV_DECIDE_FORM"] = f
            for _ in range(2):
                ctx.decide_batch_dev(dk, one.data_ptr(), m, oks.data_ptr())
            ctx.sync()
            t0 = time.perf_counter()
            for _ in range(a.reps):
                ctx.decide_batch_dev(dk, one.data_ptr(), m, oks.data_ptr())
                ctx.sync()
            res[f].append((time.perf_counter() - t0) / a.reps * 1e3)
            assert bool(oks[:m].cpu().all())
    print("decide m=%d: " % m + "  ".join("form %s %.4f ms (min %.4f)" % (f, statistics.median(v), min(v)) for f, v in res.items()), flush=True)
