"""CPU: the batched-affine PAIR LEVEL of the Pippenger (snark-verifier_amd/csrc/pair_tree.h -- the per-lane code the
k_pair_fwd / k_binv_* / k_pair_bwd kernels run) compiled for the host and executed lane by lane, against the big-integer
oracle: every pair slot of a padded bucket-sorted stream must come out as the affine sum of its two entries
(`buckets[d-1].add_assign(base)` x 2, reference snark-verifier/src/util/msm.rs:291-296), including the exceptional
pairs (P + P, P + (-P), real + pad, pad + pad) and the recursive batch inversion at several lane geometries."""
import ctypes
import random

import pytest

import bn254 as O
import coracle as C

SKIP, NEG = 1 << 30, 1 << 31


def _run(lib, pts, entries, T, m, m2, final_max):
    nslots = len(entries) // 2
    flat = (ctypes.c_uint32 * (4 * nslots))(*[w for e in entries for w in e])
    oe = (ctypes.c_uint32 * (2 * nslots))()
    oxy = ctypes.create_string_buffer(64 * nslots)
    lib.ht_pair_level.restype = ctypes.c_int
    levels = lib.ht_pair_level(b"".join(pts), len(pts), flat, nslots, T, m, m2, final_max, oe, oxy)
    return levels, list(oe), oxy.raw


def _expected(pts, e0, e1):
    def term(e):
        if e[1] & SKIP:
            return None
        p = O.g1_from_bytes(pts[e[1] & 0x3FFFFFFF])
        return O.g1_neg(p) if e[1] & NEG else p

    return O.g1_add(term(e0), term(e1))


@pytest.mark.parametrize("T,m,m2,final_max,nslots", [(4, 3, 2, 4, 157), (64, 16, 32, 1024, 3000), (8, 5, 4, 8, 1), (8, 4, 4, 8, 900), (64, 2, 2, 64, 1500)])
def test_pair_level_matches_affine_sums(hosttest_lib, T, m, m2, final_max, nslots):
    rng = random.Random(1000 * T + nslots)
    npts = 40
    raw = C.sample_points(0x9A11, npts)
    pts = [raw[64 * i:64 * i + 64] for i in range(npts)]
    entries, bucket = [], 0
    for i in range(nslots):
        kind = rng.choice(["add"] * 6 + ["dbl", "cancel", "copy", "skip", "dbl_neg", "cancel_neg"])
        if rng.random() < 0.3:
            bucket += 1
        a = rng.randrange(npts)
        b = rng.choice([x for x in range(npts) if x != a])
        sa, sb = rng.choice([0, NEG]), rng.choice([0, NEG])
        if kind == "add":
            e0, e1 = (bucket, a | sa), (bucket, b | sb)
        elif kind == "dbl":
            e0, e1 = (bucket, a | sa), (bucket, a | sa)
        elif kind == "dbl_neg":
            e0, e1 = (bucket, a | NEG), (bucket, a | NEG)
        elif kind == "cancel":
            e0, e1 = (bucket, a | sa), (bucket, a | (sa ^ NEG))
        elif kind == "cancel_neg":
            e0, e1 = (bucket, a | NEG), (bucket, a)
        elif kind == "copy":
            e0, e1 = (bucket, a | sa), (bucket, SKIP)
        else:
            e0, e1 = (bucket, SKIP), (bucket, SKIP)
        entries += [e0, e1]
    levels, oe, oxy = _run(hosttest_lib, pts, entries, T, m, m2, final_max)
    if nslots > T * m * m2:
        assert levels >= 1  # the recursion was exercised
    for i in range(nslots):
        e0, e1 = entries[2 * i], entries[2 * i + 1]
        exp = _expected(pts, e0, e1)
        assert oe[2 * i] == e0[0]                      # bucket id carried over
        assert oe[2 * i + 1] & 0x3FFFFFFF == i          # the half-length stream indexes its own point array
        assert bool(oe[2 * i + 1] & SKIP) == (exp is None), (i, e0, e1)
        assert not oe[2 * i + 1] & NEG
        assert oxy[64 * i:64 * i + 64] == O.g1_to_bytes(exp), (i, e0, e1)


def test_pair_level_equal_x_distinct_points_and_identity_free_inputs(hosttest_lib):
    """Every slot a doubling or a cancellation (the all-equal-points adversarial input of SURVEY.md 8d): no denominator
    may ever be zero -- one zero would poison the running product of its lane and, through the totals, every lane."""
    raw = C.sample_points(0x9A12, 3)
    pts = [raw[64 * i:64 * i + 64] for i in range(3)]
    entries = []
    for i in range(200):
        a = i % 3
        entries += [(i // 7, a), (i // 7, a | (NEG if i % 2 else 0))]
    _, oe, oxy = _run(hosttest_lib, pts, entries, 16, 4, 4, 16)
    for i in range(200):
        exp = _expected(pts, entries[2 * i], entries[2 * i + 1])
        assert oxy[64 * i:64 * i + 64] == O.g1_to_bytes(exp)
        assert bool(oe[2 * i + 1] & SKIP) == (exp is None)


@pytest.mark.parametrize("RUN,seed", [(4, 1), (16, 2), (64, 3), (96, 4), (2, 5)])
def test_fused_pair_runs_bucket_sums(hosttest_lib, RUN, seed):
    """The FUSED form (pairrun_fwd_lane / pairrun_bwd_lane: the backward pass adds every pair sum straight into the lane's
    bucket accumulator and leaves head / tail partials + interior buckets as k_accumulate does): every bucket of a padded
    stream must come out as the sum of its entries -- buckets of 1 .. 40 entries, so they start / end / span anywhere
    relative to the runs, incl. runs of skip slots only, doublings and cancelling pairs."""
    rng = random.Random(77 + seed)
    npts = 30
    raw = C.sample_points(0x9A20 + seed, npts)
    pts = [raw[64 * i:64 * i + 64] for i in range(npts)]
    entries, expected, nb = [], [], 60
    for b in range(nb):
        cnt = rng.choice([0, 1, 2, 3, 5, 8, 13, 40])
        real = []
        for _ in range(cnt):
            kind = rng.random()
            a = rng.randrange(npts)
            sg = rng.choice([0, NEG])
            if kind < 0.1 and real:        # repeat the previous entry: a doubling when they pair up
                real.append(real[-1])
            elif kind < 0.2 and real:      # ... or its opposite: a cancelling pair
                real.append((real[-1][0], real[-1][1] ^ NEG))
            else:
                real.append((b, a | sg))
        pad = [(b, SKIP)] * (len(real) % 2)
        if rng.random() < 0.15:
            pad += [(b, SKIP)] * (2 * rng.randrange(1, 4))  # the filler a key's last bin gets: whole skip slots
        entries += real + pad
        acc = None
        for e in real:
            acc = O.g1_add(acc, _expected(pts, e, (0, SKIP)))
        expected.append(acc)
    stop = len(entries)
    assert stop % 2 == 0
    flat = (ctypes.c_uint32 * (2 * stop))(*[w for e in entries for w in e])
    out = ctypes.create_string_buffer(64 * nb)
    hosttest_lib.ht_pair_runs.restype = ctypes.c_int
    rc = hosttest_lib.ht_pair_runs(b"".join(pts), npts, flat, stop, RUN, nb, out)
    assert rc >= 0
    for b in range(nb):
        assert out.raw[64 * b:64 * b + 64] == O.g1_to_bytes(expected[b]), (RUN, b)
