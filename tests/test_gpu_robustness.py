"""GPU: robustness of the C-ABI entry points -- random segmentations, validation
flags, concurrent contexts, determinism, adversarial duplicates at scale."""
import os
import random
import threading

import pytest

import bn254 as O
import coracle as C

pytestmark = pytest.mark.gpu


def _free_port():
    import socket

    with socket.socket() as so:
        so.bind(("127.0.0.1", 0))
        return so.getsockname()[1]


def test_batched_random_segmentations(gpu_ctx):
    rng = random.Random(21)
    n = 700
    s, p = C.sample_scalars(41, n), C.sample_points(42, n)
    for _ in range(4):
        cuts = sorted(rng.sample(range(1, n), rng.randrange(1, 60)))
        offs = [0] + cuts + [n]
        assert gpu_ctx.msm_batched(s, p, offs) == C.msm_batched(s, p, offs)
    # every segment of size 1: k*P for each term
    offs = list(range(0, 65))
    assert gpu_ctx.msm_batched(s[:32 * 64], p[:64 * 64], offs) == C.msm_batched(s[:32 * 64], p[:64 * 64], offs)


def test_pippenger_random_sizes_and_determinism(gpu_ctx):
    rng = random.Random(22)
    for _ in range(6):
        n = rng.randrange(1, 3000)
        s, p = C.sample_scalars(rng.randrange(1 << 30), n), C.sample_points(rng.randrange(1 << 30), n)
        a = gpu_ctx.msm_pippenger(s, p)
        assert a == C.msm_pippenger(s, p, 2)
        assert gpu_ctx.msm_pippenger(s, p) == a  # scratch reuse across calls changes nothing


def test_validate_flag_on_every_msm_entry_point(gpu_ctx):
    import snark_verifier_amd as sv

    n = 40
    s, p = C.sample_scalars(51, n), bytearray(C.sample_points(52, n))
    ok = gpu_ctx.msm_pippenger(s, bytes(p), flags=sv.SNARKV_FLAG_VALIDATE)
    assert ok == C.msm_pippenger(s, bytes(p), 1)
    assert gpu_ctx.msm_batched(s, bytes(p), [0, 10, n], flags=sv.SNARKV_FLAG_VALIDATE) == C.msm_batched(s, bytes(p), [0, 10, n])
    p[64 * 17 + 5] ^= 0x10  # x of point 17 perturbed: off-curve
    for call in (lambda: gpu_ctx.msm_pippenger(s, bytes(p), flags=sv.SNARKV_FLAG_VALIDATE),
                 lambda: gpu_ctx.msm_batched(s, bytes(p), [0, 10, n], flags=sv.SNARKV_FLAG_VALIDATE),
                 lambda: gpu_ctx.msm_naive(s, bytes(p), flags=sv.SNARKV_FLAG_VALIDATE)):
        with pytest.raises(sv.SnarkvError) as e:
            call()
        assert e.value.code == -3
    # coordinate >= p is rejected even when "on curve" modulo p
    q = bytearray(C.sample_points(53, 1))
    x = int.from_bytes(q[:32], "little") + O.P
    if x < (1 << 256):
        q[:32] = x.to_bytes(32, "little")
        with pytest.raises(sv.SnarkvError):
            gpu_ctx.msm_naive(C.sample_scalars(1, 1), bytes(q), flags=sv.SNARKV_FLAG_VALIDATE)


def test_duplicates_and_opposites_at_scale(gpu_ctx):
    """Many equal and opposite bases with equal scalars: the fast adders meet
    P = +-Q in a large fraction of buckets; every such bucket must be caught by
    the degeneracy test and recomputed carefully."""
    n = 4096
    base = C.sample_points(61, 8)
    pts = []
    for i in range(n):
        q = base[64 * (i % 8):64 * (i % 8) + 64]
        if (i // 8) % 3 == 2:  # every third group negated
            y = (O.P - int.from_bytes(q[32:], "little")) % O.P
            q = q[:32] + y.to_bytes(32, "little")
        pts.append(q)
    p = b"".join(pts)
    sc = [((i % 5) + 1) * 0x0123456789ABCDEF for i in range(n)]
    s = b"".join(O.fe_to_bytes(x) for x in sc)
    exp = C.msm_pippenger(s, p, 4)
    assert gpu_ctx.msm_pippenger(s, p) == exp
    assert gpu_ctx.msm_naive(s, p) == exp
    # total cancellation: P and -P with the same scalar
    half = b"".join(pts[i] for i in range(0, 16))
    neg = b"".join(q[:32] + ((O.P - int.from_bytes(q[32:], "little")) % O.P).to_bytes(32, "little") for q in
                   (half[64 * i:64 * i + 64] for i in range(16)))
    s2 = C.sample_scalars(62, 16)
    assert gpu_ctx.msm_pippenger(s2 + s2, half + neg) == b"\x00" * 64
    assert gpu_ctx.msm_naive(s2 + s2, half + neg) == b"\x00" * 64


def test_two_contexts_from_two_threads():
    """include/snarkv_amd.h "Threading": one context per host thread."""
    import snark_verifier_amd as sv

    n = 3000
    inputs = [(C.sample_scalars(70 + k, n), C.sample_points(80 + k, n)) for k in range(2)]
    expected = [C.msm_pippenger(s, p, 2) for s, p in inputs]
    results, errors = [None, None], []

    def work(k):
        try:
            ctx = sv.Context(0)
            for _ in range(3):
                results[k] = ctx.msm_pippenger(*inputs[k])
            ctx.close()
        except Exception as e:  # pragma: no cover
            errors.append(e)

    ths = [threading.Thread(target=work, args=(k,)) for k in range(2)]
    for t in ths:
        t.start()
    for t in ths:
        t.join()
    assert not errors and results == expected


def test_decide_batch_mixed_and_identity_cases(gpu_ctx, golden_decider):
    import snark_verifier_amd as sv

    g = golden_decider
    dk = sv.DecidingKey(gpu_ctx, bytes.fromhex(g["g1"]), bytes.fromhex(g["g2"]), bytes.fromhex(g["s_g2"]))
    cases = g["cases"] * 9  # 72 accumulators, accept/reject interleaved, identity cases included
    accs = b"".join(bytes.fromhex(c["acc"]) for c in cases)
    allok, oks = gpu_ctx.decide_batch(dk, accs)
    assert oks == [c["accept"] for c in cases] and not allok
    dk.close()


def test_sharded_msm_over_rccl_world_of_one_is_ordered():
    """The multi-GPU wiring (distributed.py: HIP partial -> RCCL all-gather -> HIP
    fold) on a 1-rank NCCL group, with the context on its OWN stream and inputs
    that change every call: a missing stream dependency would fold a stale or
    empty partial."""
    import os

    import torch
    import torch.distributed as dist

    import snark_verifier_amd as sv
    from snark_verifier_amd import distributed as D

    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(_free_port())  # (a fixed port can still be in TIME_WAIT from an earlier run on the box)
    dist.init_process_group("nccl", rank=0, world_size=1, device_id=torch.device("cuda", 0))
    try:
        ctx = sv.Context(0)  # private stream
        ref = sv.Context(0)
        n = 1 << 14
        ds = torch.empty(32 * n, dtype=torch.uint8, device="cuda")
        dp = torch.empty(64 * n, dtype=torch.uint8, device="cuda")
        exp = torch.zeros(64, dtype=torch.uint8, device="cuda")
        seen = set()
        for it in range(6):
            ref.sample_scalars_dev(1000 + it, n, ds.data_ptr())
            ref.sample_points_dev(2000 + it, n, dp.data_ptr())
            ref.msm_pippenger_dev(ds.data_ptr(), dp.data_ptr(), n, exp.data_ptr(), 0)
            ref.sync()
            # force the real collective even at world size 1
            part = torch.zeros(sv.G1_PARTIAL_BYTES, dtype=torch.uint8, device="cuda")
            gathered = torch.zeros(sv.G1_PARTIAL_BYTES, dtype=torch.uint8, device="cuda")
            out = torch.zeros(64, dtype=torch.uint8, device="cuda")
            ctx.msm_pippenger_partial_dev(ds.data_ptr(), dp.data_ptr(), n, part.data_ptr(), 0)
            ctx.sync()
            dist.all_gather_into_tensor(gathered, part)
            torch.cuda.current_stream().synchronize()
            ctx.fold_partials_dev(gathered.data_ptr(), 1, out.data_ptr())
            ctx.sync()
            assert bytes(out.cpu().numpy()) == bytes(exp.cpu().numpy())
            got = D.gpu_sharded_msm(ctx, ds, dp, n)
            ctx.sync()
            assert bytes(got.cpu().numpy()) == bytes(exp.cpu().numpy())
            seen.add(bytes(exp.cpu().numpy()))
        assert len(seen) == 6
    finally:
        dist.destroy_process_group()


def test_sharded_msm_batch_wiring_on_a_one_rank_rccl_group():
    """`gpu_sharded_msm_batch` (K partials -> ONE RCCL all-gather -> K folds in one launch; what `bench.py --gpus N`
    times) on a 1-rank NCCL group, context on a torch side stream, inputs that change every call: every job must equal
    the single-call result of the same inputs."""
    import os

    import torch
    import torch.distributed as dist

    import snark_verifier_amd as sv
    from snark_verifier_amd import distributed as D

    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(_free_port())
    dist.init_process_group("nccl", rank=0, world_size=1, device_id=torch.device("cuda", 0))
    try:
        side = torch.cuda.Stream()
        ctx = sv.Context(0, stream=side.cuda_stream)
        ref = sv.Context(0)
        sizes = [1 << 13, 5000, 1 << 13, 777]
        ds = [torch.empty(32 * n, dtype=torch.uint8, device="cuda") for n in sizes]
        dp = [torch.empty(64 * n, dtype=torch.uint8, device="cuda") for n in sizes]
        exp = torch.zeros(64 * len(sizes), dtype=torch.uint8, device="cuda")
        for it in range(4):
            for i, n in enumerate(sizes):
                ref.sample_scalars_dev(3000 + 10 * it + i, n, ds[i].data_ptr())
                ref.sample_points_dev(4000 + 10 * it + i, n, dp[i].data_ptr())
                ref.msm_pippenger_dev(ds[i].data_ptr(), dp[i].data_ptr(), n, exp.data_ptr() + 64 * i, 0)
            ref.sync()
            torch.cuda.synchronize()
            got = D.gpu_sharded_msm_batch(ctx, ds, dp, sizes, stream=side)
            side.synchronize()
            assert bytes(got.cpu().numpy()) == bytes(exp.cpu().numpy()), it
    finally:
        dist.destroy_process_group()


def test_context_free_entry_points_from_many_threads(golden_msm):
    """`bn254_*` draw from the pool of default contexts (csrc/capi.hip): concurrent callers run side by side, no races."""
    import ctypes
    import threading

    import snark_verifier_amd as sv

    lib = sv.load_library()
    cases = [c for c in golden_msm if len(c["scalars"]) // 64 >= 2][:6]
    errs = []

    def worker(k):
        for it in range(12):
            c = cases[(k + it) % len(cases)]
            s, p = bytes.fromhex(c["scalars"]), bytes.fromhex(c["points"])
            out = ctypes.create_string_buffer(64)
            fn = lib.bn254_g1_msm_naive if (k + it) % 2 else lib.bn254_g1_msm_pippenger
            rc = fn(s, p, len(s) // 32, out)
            if rc != 0 or out.raw != bytes.fromhex(c["expected"]):
                errs.append((k, it, rc))

    ts = [threading.Thread(target=worker, args=(k,)) for k in range(8)]
    for t in ts:
        t.start()
    for t in ts:
        t.join()
    assert not errs


def test_bucket_sharded_msm_world1_rccl_wiring():
    """`gpu_bucket_sharded_msm` end to end over a 1-rank RCCL group (the 8-GPU run is the
    driver's): fill -> (no exchange at world 1) -> reduce -> fold equals the plain MSM."""
    import torch
    import torch.distributed as dist

    import snark_verifier_amd as sv
    from snark_verifier_amd.distributed import gpu_bucket_sharded_msm

    created = False
    if not dist.is_initialized():
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        os.environ["MASTER_PORT"] = str(_free_port())
        dist.init_process_group("nccl", rank=0, world_size=1)
        created = True
    try:
        n = 30000
        st = torch.cuda.Stream()
        ctx = sv.Context(0, st.cuda_stream)
        ds = torch.empty(32 * n, dtype=torch.uint8, device="cuda")
        dp = torch.empty(64 * n, dtype=torch.uint8, device="cuda")
        torch.cuda.synchronize()
        ctx.sample_scalars_dev(3, n, ds.data_ptr())
        ctx.sample_points_dev(4, n, dp.data_ptr())
        ref = torch.zeros(64, dtype=torch.uint8, device="cuda")
        torch.cuda.synchronize()  # (the zero fill runs on torch's stream, the context on `st`)
        ctx.msm_pippenger_dev(ds.data_ptr(), dp.data_ptr(), n, ref.data_ptr())
        ctx.sync()
        out = gpu_bucket_sharded_msm(ctx, ds, dp, n)
        ctx.sync()
        assert bytes(out.cpu().numpy()) == bytes(ref.cpu().numpy()) != bytes(64)
    finally:
        if created:
            dist.destroy_process_group()


def test_empty_shard_contributes_the_identity(gpu_ctx):
    """n < world: a rank whose shard is empty must hand the all-gather the identity partial instead of
    raising on its own while the peers block.  Emulated here: 5 points over 8 ranks -- ranks 5..7 are empty -- through
    `distributed.gpu_msm_partial` (the per-rank step of `gpu_sharded_msm`) and the device fold."""
    import torch

    import coracle as C
    import snark_verifier_amd as sv
    from snark_verifier_amd.distributed import gpu_msm_partial, shard_range

    n, world = 5, 8
    s, p = C.sample_scalars(0x91, n), C.sample_points(0x92, n)
    ds = torch.frombuffer(bytearray(s), dtype=torch.uint8).cuda()
    dp = torch.frombuffer(bytearray(p), dtype=torch.uint8).cuda()
    gathered = torch.zeros(world, sv.G1_PARTIAL_BYTES, dtype=torch.uint8, device="cuda")
    empties = 0
    for r in range(world):
        lo, hi = shard_range(n, r, world)
        empties += hi == lo
        part = torch.full((sv.G1_PARTIAL_BYTES,), 0xAB, dtype=torch.uint8, device="cuda")  # stale bytes must not survive
        gpu_msm_partial(gpu_ctx, part, ds[32 * lo:], dp[64 * lo:], hi - lo)
        gathered[r] = part
    assert empties == 3
    out = torch.zeros(64, dtype=torch.uint8, device="cuda")
    torch.cuda.synchronize()
    gpu_ctx.fold_partials_dev(gathered.data_ptr(), world, out.data_ptr())
    gpu_ctx.sync()
    got, want = bytes(out.cpu().numpy()), C.msm_pippenger(s, p, 1)
    if got != want:  # (round 5: red once on the driver's box, never reproduced -- profiles/r06_flaky_empty_shard.txt; say WHICH rank)
        raw = bytes(gathered.cpu().numpy())
        report = []
        for r in range(world):
            one = torch.zeros(64, dtype=torch.uint8, device="cuda")
            gpu_ctx.fold_partials_dev(gathered[r].data_ptr(), 1, one.data_ptr())
            lo, hi = shard_range(n, r, world)
            exp_r = C.msm_pippenger(s[32 * lo:32 * hi], p[64 * lo:64 * hi], 1) if hi > lo else bytes(64)
            report.append("rank %d %s: %s" % (r, "ok" if bytes(one.cpu().numpy()) == exp_r else "WRONG", raw[144 * r:144 * r + 144].hex()))
        raise AssertionError("fold %s != oracle %s\n%s" % (got.hex(), want.hex(), "\n".join(report)))


def test_inputs_in_pinned_host_buffers(gpu_ctx):
    """`snarkv_ctx_host_buffer`: inputs assembled in the context's pinned memory give the same bytes as pageable
    ones, the slots are independent, and a slot that grows stays usable."""
    import ctypes

    for n in (50, 3000):  # the second request outgrows the first: the slots are reallocated
        s, p = C.sample_scalars(61 + n, n), C.sample_points(62 + n, n)
        hs, hp = gpu_ctx.host_buffer(0, 32 * n), gpu_ctx.host_buffer(1, 64 * n)
        assert ctypes.addressof(hs) != ctypes.addressof(hp)
        ctypes.memmove(hs, s, len(s))
        ctypes.memmove(hp, p, len(p))
        offs = [0, 7, n // 2, n]
        want = C.msm_batched(s, p, offs)
        assert gpu_ctx.msm_batched(hs, hp, offs) == want == gpu_ctx.msm_batched(s, p, offs)
        assert gpu_ctx.msm_pippenger(hs, hp) == C.msm_pippenger(s, p, 2)
    again = gpu_ctx.host_buffer(0, 64)  # a smaller request returns the same (grown) slot
    assert ctypes.addressof(again) == ctypes.addressof(gpu_ctx.host_buffer(0, 32 * 3000))
    import snark_verifier_amd as sv

    with pytest.raises(sv.SnarkvError):
        gpu_ctx.host_buffer(sv.SNARKV_HOST_BUFFERS, 16)


def test_joint_fixed_window_form_of_large_segmented_launches(gpu_ctx):
    """Above 16 384 terms a segmented launch multiplies with the fixed-window kernel; with the context's throughput hint
    (or from 49 152 terms) it takes the GROUP form -- K <= 4 terms of a segment per lane, all their GLV halves on shared
    doublings (k_term_scalar_mul_group).  Ragged segments, scalars that stress the two digit streams (0, 1, r - 1, lambda and small
    combinations a + b lambda whose halves are tiny, 2^k, all-ones), repeated and opposite points, identities."""
    rng = random.Random(77)
    n = 20000
    s = bytearray(C.sample_scalars(81, n))
    p = bytearray(C.sample_points(82, n))
    lam = 0xb3c4d79d41a917585bfc41088d8daaa78b17ea66b99c90dd  # a cube root of unity mod r (glv.h)
    assert pow(lam, 3, O.R) == 1 and lam != 1
    special = [0, 1, 2, 7, 8, 9, O.R - 1, O.R - 2, lam, (lam * lam) % O.R, (1 + lam) % O.R, (3 + 4 * lam) % O.R,
               (O.R - 5 * lam) % O.R, (8 * lam + 1) % O.R, (1 << 126) % O.R, (1 << 127) % O.R, (1 << 253) % O.R, (1 << 254) - 1 - O.R]
    special += [(1 << k) for k in range(0, 250, 17)]
    for i, v in enumerate(special):
        s[32 * (100 + i):32 * (101 + i)] = (v % O.R).to_bytes(32, "little")
    # repeated / opposite points inside one segment, and identities
    for i in range(40):
        p[64 * (300 + i):64 * (301 + i)] = p[64 * 300:64 * 301]
    x = int.from_bytes(p[64 * 350:64 * 350 + 32], "little")
    y = int.from_bytes(p[64 * 350 + 32:64 * 351], "little")
    p[64 * 351:64 * 352] = x.to_bytes(32, "little") + (O.P - y).to_bytes(32, "little")
    s[32 * 351:32 * 352] = s[32 * 350:32 * 351]  # k P + k (-P) = O
    for i in (400, 401, 777):
        p[64 * i:64 * (i + 1)] = bytes(64)
    cuts = sorted(rng.sample(range(1, n), 900))
    offs = [0] + cuts + [n]
    offs = [o for o in offs if not (349 < o <= 351)]  # keep the opposite pair in one segment
    want = C.msm_batched(bytes(s), bytes(p), offs)
    gpu_ctx.set_throughput_hint(True)
    try:
        got = gpu_ctx.msm_batched(bytes(s), bytes(p), offs)
    finally:
        gpu_ctx.set_throughput_hint(False)
    assert got == want
    assert gpu_ctx.msm_batched(bytes(s), bytes(p), offs) == want  # and the two-lane form on the same input
