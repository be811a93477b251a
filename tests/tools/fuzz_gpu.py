"""Dev tool: randomized differential run of the C-ABI MSM entry points against the C oracle
(random sizes, special scalars, identity / repeated / opposite bases, random segmentations)."""
import ctypes, os, random, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "oracle"))
import bn254 as O
import coracle as C
import snark_verifier_amd as sv

secs = float(sys.argv[1]) if len(sys.argv) > 1 else 60
rng = random.Random(int(sys.argv[2]) if len(sys.argv) > 2 else 1)
import torch
ctx = sv.Context(0)
lib = sv.load_library()
mgs = {}
R = O.R
special = [0, 1, 2, R - 1, R - 2, (R - 1) // 2, 1 << 127, (1 << 127) - 1, 1 << 128, (1 << 253), 0xFFFFFFFF, 1 << 64]
t0 = time.time(); cases = 0; terms = 0
while time.time() - t0 < secs:
    n = rng.choice([1, 2, 3, 5, 31, 32, 33, 64, 65, 100, 255, 256, 1000, 4095, 4096, 4097, rng.randrange(1, 20000), rng.randrange(1, 3000)])
    seed = rng.randrange(1 << 30)
    s = bytearray(C.sample_scalars(seed, n)); p = bytearray(C.sample_points(seed + 1, n))
    for _ in range(rng.randrange(0, 6)):   # sprinkle specials
        i = rng.randrange(n); s[32 * i:32 * i + 32] = rng.choice(special).to_bytes(32, "little")
    for _ in range(rng.randrange(0, 4)):
        i = rng.randrange(n); kind = rng.randrange(3)
        if kind == 0: p[64 * i:64 * i + 64] = bytes(64)
        elif n > 1:
            j = rng.randrange(n); pj = bytes(p[64 * j:64 * j + 64])
            if kind == 1: p[64 * i:64 * i + 64] = pj
            else:
                q = O.g1_from_bytes(pj); p[64 * i:64 * i + 64] = O.g1_to_bytes(O.g1_neg(q)) if q else pj
                if rng.random() < 0.5: s[32 * i:32 * i + 32] = s[32 * j:32 * j + 32]
    s, p = bytes(s), bytes(p)
    exp = C.msm_pippenger(s, p, 8)
    assert ctx.msm_pippenger(s, p) == exp, ("pippenger", n, seed)
    if n <= 6000:
        assert ctx.msm_naive(s, p) == exp, ("naive", n, seed)
        k = rng.randrange(1, min(n, 40) + 1)
        cuts = sorted(rng.sample(range(1, n), k - 1)) if k > 1 else []
        offs = [0] + cuts + [n]
        assert ctx.msm_batched(s, p, offs) == C.msm_batched(s, p, offs), ("batched", n, seed, offs)
    # round 5: the same inputs through the context-free boundary (a pool context per call) ...
    out = ctypes.create_string_buffer(64)
    assert lib.bn254_g1_msm_pippenger(s, p, n, out) == 0 and out.raw == exp, ("bn254 pippenger", n, seed)
    # ... and, every few cases, sharded over 1-4 emulated ranks as a BATCH of two jobs through the single-process multi-GPU
    # entry point (job 1 = the same points with the scalars reversed)
    if cases % 5 == 0 and n <= 6000:
        world = rng.randrange(1, 5)
        mg = mgs.setdefault(world, sv.MultiGpu([0] * world))
        s2 = b"".join(s[32 * i:32 * i + 32] for i in reversed(range(n)))
        keep, ds, dp, cn = [], [], [], []
        for g in range(world):
            lo, hi = mg.shard(n, g)
            row_s, row_p, row_n = [], [], []
            for sj in (s, s2):
                if hi > lo:
                    ts = torch.frombuffer(bytearray(sj[32 * lo:32 * hi]), dtype=torch.uint8).cuda()
                    tp = torch.frombuffer(bytearray(p[64 * lo:64 * hi]), dtype=torch.uint8).cuda()
                    keep += [ts, tp]
                    row_s.append(ts.data_ptr()), row_p.append(tp.data_ptr())
                else:
                    row_s.append(None), row_p.append(None)
                row_n.append(hi - lo)
            ds.append(row_s), dp.append(row_p), cn.append(row_n)
        torch.cuda.synchronize()
        assert mg.msm_pippenger_many_dev(ds, dp, cn) == [exp, C.msm_pippenger(s2, p, 8)], ("mgpu batch", n, seed, world)
    cases += 1; terms += n
print("fuzz ok: %d cases, %d terms, %.0f s" % (cases, terms, time.time() - t0))
