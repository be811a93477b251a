"""Dev tool: context churn + random sizes for a fixed wall time; reports device
memory before/after (leak check) and verifies determinism of every repeated call."""
import os, random, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
import snark_verifier_amd as sv

secs = float(sys.argv[1]) if len(sys.argv) > 1 else 30
rng = random.Random(1)
free0 = torch.cuda.mem_get_info()[0]
base = sv.Context(0)
N = 1 << 18
ds = torch.empty(32 * N, dtype=torch.uint8, device="cuda"); dp = torch.empty(64 * N, dtype=torch.uint8, device="cuda")
base.sample_scalars_dev(7, N, ds.data_ptr()); base.sample_points_dev(8, N, dp.data_ptr()); base.sync()
g2 = bytes.fromhex("edf692d95cbdde46ddda5ef7d422436779445c5e66006a42761e1f12efde0018c212f3aeb785e49712e7a9353349aaf1255dfb31b7bf60723a480d9293938e19"
                   "aa7dfa6601cce64c7bd3430c69e7d1e38f40cb8d8071ab4aeb6d8cdba55ec8125b9722d1dcdaac55f38eb37033314bbc95330c69ad999eec75f05f58d0890609")
g1 = (1).to_bytes(32, "little") + (2).to_bytes(32, "little")
cache = {}
t0 = time.time(); it = 0
while time.time() - t0 < secs:
    ctx = sv.Context(0)
    dk = sv.DecidingKey(ctx, g1, g2, g2)
    for _ in range(rng.randrange(1, 6)):
        n = rng.choice([1, 2, 3, 33, 1000, 4097, 1 << 14, 1 << 16, 1 << 18, rng.randrange(1, 1 << 17)])
        c = rng.choice([0, 0, 8, 12, 16])
        out = torch.zeros(64, dtype=torch.uint8, device="cuda")
        ctx.msm_pippenger_dev(ds.data_ptr(), dp.data_ptr(), n, out.data_ptr(), c)
        ctx.sync()
        h = bytes(out.cpu().numpy())
        assert cache.setdefault(("p", n), h) == h, ("pippenger nondeterministic", n, c)
        if n <= 1 << 14:
            k = rng.randrange(1, min(n, 50) + 1)
            cuts = sorted(rng.sample(range(1, n), k - 1)) if n > 1 and k > 1 else []
            offs = [0] + cuts + [n]
            o = torch.tensor(offs, dtype=torch.int32, device="cuda")
            ob = torch.zeros(64 * (len(offs) - 1), dtype=torch.uint8, device="cuda")
            ctx.msm_batched_dev(ds.data_ptr(), dp.data_ptr(), o.data_ptr(), len(offs) - 1, n, ob.data_ptr())
            ctx.sync()
        m = rng.choice([1, 2, 17, 513, 600])
        accs = torch.frombuffer(bytearray((g1 + g1) * m), dtype=torch.uint8).cuda()
        oks = torch.zeros(m, dtype=torch.uint8, device="cuda")
        ctx.decide_batch_dev(dk, accs.data_ptr(), m, oks.data_ptr()); ctx.sync()
        assert bool(oks.cpu().all())
    dk.close(); ctx.close(); it += 1
torch.cuda.synchronize(); torch.cuda.empty_cache()
free1 = torch.cuda.mem_get_info()[0]
print("iterations", it, "free before %.1f MiB after %.1f MiB (delta %.1f MiB)" % (free0 / 2**20, free1 / 2**20, (free0 - free1) / 2**20))
