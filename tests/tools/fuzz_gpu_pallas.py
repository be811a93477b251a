"""Dev tool: randomized differential run of the pasta build's MSM entry points (large, small, segmented)
against the pure-Python pallas oracle: special scalars (0, 1, r-1, +-lambda multiples, powers of two around
the GLV split), identity / repeated / opposite bases, random segmentations.
Run on the GPU box: python tests/tools/fuzz_gpu_pallas.py [seconds] [seed]"""
import os
import random
import sys
import time

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "oracle"))
import pallas as PA
from snark_verifier_amd import pallas as PL

secs = float(sys.argv[1]) if len(sys.argv) > 1 else 60
rng = random.Random(int(sys.argv[2]) if len(sys.argv) > 2 else 1)
ctx = PL.PallasContext(0)
R = PA.R
lam = next(w for w in (pow(g, (R - 1) // 3, R) for g in range(2, 20)) if w != 1)
special = [0, 1, 2, R - 1, R - 2, (R - 1) // 2, lam, R - lam, lam * lam % R, (lam + 1) % R, 1 << 126, (1 << 127) - 1, 1 << 127,
           (1 << 128) - 1, 1 << 253, (1 << 254) % R, 0xFFFFFFFF, 1 << 64]
pool = PA.sample_points(99, 400)
enc_s = lambda v: b"".join(PA.fe_to_bytes(x) for x in v)  # noqa: E731
enc_p = lambda v: b"".join(PA.g1_to_bytes(x) for x in v)  # noqa: E731
t0 = time.time()
cases = terms = 0
while time.time() - t0 < secs:
    n = rng.choice([1, 2, 3, 5, 31, 32, 33, 64, 65, 100, 255, 256, rng.randrange(1, 700)])
    sc = [rng.randrange(R) for _ in range(n)]
    pts = [rng.choice(pool) for _ in range(n)]
    for _ in range(rng.randrange(0, 6)):
        sc[rng.randrange(n)] = rng.choice(special)
    for _ in range(rng.randrange(0, 4)):
        i, kind = rng.randrange(n), rng.randrange(3)
        if kind == 0:
            pts[i] = None
        else:
            j = rng.randrange(n)
            if pts[j] is not None:
                pts[i] = pts[j] if kind == 1 else PA.g1_neg(pts[j])
                if rng.random() < 0.5:
                    sc[i] = sc[j]
    live = [(s, p) for s, p in zip(sc, pts) if p is not None]
    want = PA.g1_msm_pippenger([s for s, _ in live], [p for _, p in live]) if live else None
    s, p = enc_s(sc), enc_p(pts)
    got = ctx.msm_pippenger(s, p)
    assert got == PA.g1_to_bytes(want), ("pippenger", n, cases)
    if n <= 300:
        assert ctx.msm_naive(s, p) == PA.g1_to_bytes(want), ("naive", n, cases)
        cuts = sorted(set([0, n] + [rng.randrange(1, n) for _ in range(rng.randrange(0, 4)) if n > 1]))
        segs = ctx.msm_batched(s, p, cuts)
        for a, (lo, hi) in enumerate(zip(cuts, cuts[1:])):
            lv = [(x, q) for x, q in zip(sc[lo:hi], pts[lo:hi]) if q is not None]
            w = PA.g1_msm_pippenger([x for x, _ in lv], [q for _, q in lv]) if lv else None
            assert segs[64 * a:64 * a + 64] == PA.g1_to_bytes(w), ("batched", n, cases, a)
    cases += 1
    terms += n
print("pallas fuzz ok: %d cases, %d terms, %.0f s" % (cases, terms, time.time() - t0))
