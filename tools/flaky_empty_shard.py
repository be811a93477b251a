"""Root-cause experiment for the round-5 red test `test_empty_shard_contributes_the_identity`.

The driver's fresh box returned b'~7\\x8c\\x809' ... b'6\\x9b\\xf1\\x16' where the oracle says b'`i\\x1e\\xa3' ... -- not any
small-coefficient combination of the five one-point partials (checked on the CPU with the oracle), so some partial held
bytes that are no group element.  This script runs, on the GPU box:

  legacy      the round-5 wiring (kept HERE only, to reproduce): `torch.full(0xAB)` on torch's stream, then the context on
              its private non-blocking stream writes the same 144 bytes with no ordering, `ctx.sync()`, torch copies
  legacy+sync `torch.cuda.synchronize()` between the fill and the context's launch
  legacy+zero the 0xAB fill replaced by zeros
  events      the round-6 wiring (`snarkv_ctx_wait_stream` / `snarkv_stream_wait_ctx`, distributed.gpu_msm_partial)
  each alone (ITERS times) and directly after `gpu_bucket_sharded_msm` (the test that preceded it in the suite), plus
  legacy+ballast  a 256 MiB torch fill in front of the 0xAB fill: delays torch's stream the way a busy box would
  poison r    the suspected outcome made deterministic: rank r's partial overwritten with 0xAB AFTER the context wrote
              it; the fold's bytes are compared with the driver's failure bytes

Also times the world-1 step (`gpu_sharded_msm`, n = 5) with host syncs (legacy) against events.
Usage: python tools/flaky_empty_shard.py [iters] > profiles/r06_flaky_empty_shard.txt"""
import os
import sys
import time

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "oracle"))

import torch  # noqa: E402

import coracle as C  # noqa: E402
import snark_verifier_amd as sv  # noqa: E402
from snark_verifier_amd import distributed as D  # noqa: E402

ITERS = int(sys.argv[1]) if len(sys.argv) > 1 else 500
FAIL_HEAD, FAIL_TAIL = bytes([0x7E, 0x37, 0x8C, 0x80, 0x39]), bytes([0x36, 0x9B, 0xF1, 0x16])
n, world = 5, 8
s, p = C.sample_scalars(0x91, n), C.sample_points(0x92, n)
want = C.msm_pippenger(s, p, 1)
ds = torch.frombuffer(bytearray(s), dtype=torch.uint8).cuda()
dp = torch.frombuffer(bytearray(p), dtype=torch.uint8).cuda()
ballast = torch.empty(256 << 20, dtype=torch.uint8, device="cuda")
torch.cuda.synchronize()


def legacy_partial(ctx, part, d_s, d_p, count):
    if count == 0:
        part.zero_()
        torch.cuda.current_stream().synchronize()
        return
    ctx.msm_pippenger_partial_dev(d_s.data_ptr(), d_p.data_ptr(), count, part.data_ptr(), 0)
    ctx.sync()


def one_round(ctx, variant, poison=-1):
    gathered = torch.zeros(world, sv.G1_PARTIAL_BYTES, dtype=torch.uint8, device="cuda")
    for r in range(world):
        lo, hi = D.shard_range(n, r, world)
        if variant == "legacy+ballast":
            ballast.fill_(r)
        fill = 0 if variant == "legacy+zero" else 0xAB
        part = torch.full((sv.G1_PARTIAL_BYTES,), fill, dtype=torch.uint8, device="cuda")
        if variant == "legacy+sync":
            torch.cuda.synchronize()
        if variant == "events":
            D.gpu_msm_partial(ctx, part, ds[32 * lo:], dp[64 * lo:], hi - lo)
        else:
            legacy_partial(ctx, part, ds[32 * lo:], dp[64 * lo:], hi - lo)
        if r == poison:
            torch.cuda.synchronize()
            part.fill_(0xAB)
        gathered[r] = part
    out = torch.zeros(64, dtype=torch.uint8, device="cuda")
    torch.cuda.synchronize()
    ctx.fold_partials_dev(gathered.data_ptr(), world, out.data_ptr())
    ctx.sync()
    return bytes(out.cpu().numpy())


def bucket_sharded_first(ctx):
    m = 30000
    a = torch.empty(32 * m, dtype=torch.uint8, device="cuda")
    b = torch.empty(64 * m, dtype=torch.uint8, device="cuda")
    torch.cuda.synchronize()
    ctx.sample_scalars_dev(3, m, a.data_ptr())
    ctx.sample_points_dev(4, m, b.data_ptr())
    ctx.sync()
    D.gpu_bucket_sharded_msm(ctx, a, b, m)
    torch.cuda.synchronize()


print("device:", torch.cuda.get_device_name(0), "| iters per variant:", ITERS)
print("oracle bytes      :", want[:5].hex(), "...", want[-4:].hex())
print("driver's r05 bytes:", FAIL_HEAD.hex(), "...", FAIL_TAIL.hex())
ctx = sv.Context(0, ordered=False)  # the r05 Context: a private stream and NO automatic ordering
print("\n-- each variant alone, %d rounds (8 emulated ranks, fold, compare with the oracle) --" % ITERS)
for variant in ("legacy", "legacy+sync", "legacy+zero", "events", "legacy+ballast"):
    bad, seen_driver = 0, 0
    for it in range(ITERS):
        got = one_round(ctx, variant)
        bad += got != want
        seen_driver += got[:5] == FAIL_HEAD and got[-4:] == FAIL_TAIL
    print("%-15s mismatches %4d / %d   (equal to the driver's failure bytes: %d)" % (variant, bad, ITERS, seen_driver))
print("\n-- directly after gpu_bucket_sharded_msm (the preceding test of the suite), 50 times each --")
for variant in ("legacy", "events"):
    bad = 0
    for it in range(50):
        bucket_sharded_first(ctx)
        bad += one_round(ctx, variant) != want
    print("%-15s mismatches %4d / 50" % (variant, bad))
print("\n-- directly after the LIFE of a 1-rank RCCL group (init_process_group(nccl) + gpu_bucket_sharded_msm + destroy: what")
print("   test_bucket_sharded_msm_world1_rccl_wiring did right before the red test), 30 times each --")
import torch.distributed as dist  # noqa: E402


def rccl_group_life(ctx):
    import socket

    with socket.socket() as so:
        so.bind(("127.0.0.1", 0))
        port = so.getsockname()[1]
    os.environ["MASTER_ADDR"], os.environ["MASTER_PORT"] = "127.0.0.1", str(port)
    dist.init_process_group("nccl", rank=0, world_size=1)
    try:
        bucket_sharded_first(ctx)
    finally:
        dist.destroy_process_group()


for variant in ("legacy", "events"):
    bad, seen_driver, t_stall = 0, 0, []
    for it in range(30):
        rccl_group_life(ctx)
        t0 = time.perf_counter()
        torch.zeros(1, device="cuda")  # how long torch's stream is held up after the group's teardown
        got = one_round(ctx, variant)
        bad += got != want
        seen_driver += got[:5] == FAIL_HEAD and got[-4:] == FAIL_TAIL
    print("%-15s mismatches %4d / 30   (equal to the driver's failure bytes: %d)" % (variant, bad, seen_driver))

print("\n-- the suspected outcome made deterministic: rank r's partial = 0xAB bytes after the context wrote it --")
for r in range(5):
    got = one_round(ctx, "legacy+sync", poison=r)
    print("poison rank %d -> %s ... %s   %s" % (r, got[:5].hex(), got[-4:].hex(),
                                              "== the driver's failure bytes" if got[:5] == FAIL_HEAD and got[-4:] == FAIL_TAIL else ""))

print("\n-- ... and for a PREFIX of ranks (torch's stream held up while the context ran ahead: its fills of ranks 0 .. k land late) --")


def one_round_poison_set(ctx, ranks):
    gathered = torch.zeros(world, sv.G1_PARTIAL_BYTES, dtype=torch.uint8, device="cuda")
    for r in range(world):
        lo, hi = D.shard_range(n, r, world)
        part = torch.zeros(sv.G1_PARTIAL_BYTES, dtype=torch.uint8, device="cuda")
        torch.cuda.synchronize()
        legacy_partial(ctx, part, ds[32 * lo:], dp[64 * lo:], hi - lo)
        torch.cuda.synchronize()
        if r in ranks:
            part.fill_(0xAB)
        gathered[r] = part
    out = torch.zeros(64, dtype=torch.uint8, device="cuda")
    torch.cuda.synchronize()
    ctx.fold_partials_dev(gathered.data_ptr(), world, out.data_ptr())
    ctx.sync()
    return bytes(out.cpu().numpy())


import itertools  # noqa: E402

hits = []
for k in range(1, 6):
    for ranks in itertools.combinations(range(5), k):
        got = one_round_poison_set(ctx, set(ranks))
        hit = got[:5] == FAIL_HEAD and got[-4:] == FAIL_TAIL
        if hit:
            hits.append(ranks)
        if hit or ranks == tuple(range(k)):
            print("poison ranks %-16s -> %s ... %s   %s" % (ranks, got[:5].hex(), got[-4:].hex(), "== the driver's failure bytes" if hit else ""))
print("poison sets equal to the driver's failure bytes:", hits or "none of the 31 subsets of ranks 0..4")

print("\n-- (c) the per-rank partials themselves: rank r's 144 bytes folded alone == the oracle's s_r * P_r --")
real = []
for r in range(5):
    part = torch.zeros(sv.G1_PARTIAL_BYTES, dtype=torch.uint8, device="cuda")
    o1 = torch.zeros(64, dtype=torch.uint8, device="cuda")
    torch.cuda.synchronize()
    legacy_partial(ctx, part, ds[32 * r:], dp[64 * r:], 1)
    ctx.fold_partials_dev(part.data_ptr(), 1, o1.data_ptr())
    ctx.sync()
    real.append(bytes(part.cpu().numpy()))
    ok = bytes(o1.cpu().numpy()) == C.msm_pippenger(s[32 * r:32 * r + 32], p[64 * r:64 * r + 64], 1)
    print("rank %d: partial -> affine %s the oracle" % (r, "==" if ok else "!="))

print("\n-- a fill landing BETWEEN the context's stores of one partial: every 4-byte split of one rank's 144 bytes, 0xAB on either side --")


def fold_bytes(parts):
    g = torch.frombuffer(bytearray(b"".join(parts)), dtype=torch.uint8).cuda()
    o1 = torch.zeros(64, dtype=torch.uint8, device="cuda")
    torch.cuda.synchronize()
    ctx.fold_partials_dev(g.data_ptr(), len(parts), o1.data_ptr())
    ctx.sync()
    return bytes(o1.cpu().numpy())


assert fold_bytes(real + [bytes(144)] * 3) == want
split_hits, tried = [], 0
for r in range(5):
    for j in range(4, 144, 4):
        for side in (0, 1):
            mixed = (b"\xab" * j + real[r][j:]) if side == 0 else (real[r][:j] + b"\xab" * (144 - j))
            got = fold_bytes(real[:r] + [mixed] + real[r + 1:] + [bytes(144)] * 3)
            tried += 1
            if got[:5] == FAIL_HEAD and got[-4:] == FAIL_TAIL:
                split_hits.append((r, j, side))
print("split overwrites equal to the driver's failure bytes: %s of %d tried" % (split_hits or "none", tried))

print("\n-- every mixture: gathered[r] in {its own partial, any other rank's partial, 0xAB bytes, zeros} for r = 0..4 (7^5 folds in one launch) --")
opts = real + [b"\xab" * 144, bytes(144)]
combos = list(itertools.product(range(7), repeat=5))
blob = bytearray()
for cmb in combos:
    for r, o in enumerate(cmb):
        blob += opts[o]
    blob += bytes(144) * 3
gm = torch.frombuffer(blob, dtype=torch.uint8).cuda()
om = torch.zeros(64 * len(combos), dtype=torch.uint8, device="cuda")
torch.cuda.synchronize()
ctx.fold_partials_many_dev(gm.data_ptr(), 8, len(combos), om.data_ptr())
ctx.sync()
res = bytes(om.cpu().numpy())
mix_hits = [combos[i] for i in range(len(combos)) if res[64 * i:64 * i + 5] == FAIL_HEAD and res[64 * i + 60:64 * i + 64] == FAIL_TAIL]
ident = combos.index((0, 1, 2, 3, 4))
assert res[64 * ident:64 * ident + 64] == want
print("mixtures equal to the driver's failure bytes: %s of %d (index: 0-4 = rank's partial, 5 = 0xAB, 6 = zeros)" % (mix_hits or "none", len(combos)))

print("\n-- mechanism check: the loop of tests/test_gpu_stream_order.py WITHOUT any ordering (expected to FAIL: it shows the tests can) --")
nn = 2048
sets = []
for i in range(3):
    a, b = C.sample_scalars(0x600 + 2 * i, nn), C.sample_points(0x601 + 2 * i, nn)
    sets.append((torch.frombuffer(bytearray(a), dtype=torch.uint8).cuda(), torch.frombuffer(bytearray(b), dtype=torch.uint8).cuda(), C.msm_pippenger(a, b, 2)))
d_s = torch.empty(32 * nn, dtype=torch.uint8, device="cuda")
d_p = torch.empty(64 * nn, dtype=torch.uint8, device="cuda")
for mode in ("no ordering", "wait_stream only", "both (the product wiring)"):
    kept = torch.zeros(200, 64, dtype=torch.uint8, device="cuda")
    torch.cuda.synchronize()
    for it in range(200):
        a, b, _ = sets[it % 3]
        ballast.fill_(it & 0xFF)
        d_s.copy_(a)
        d_p.copy_(b)
        o = torch.full((64,), 0xAB, dtype=torch.uint8, device="cuda")
        if mode != "no ordering":
            ctx.wait_stream()
        ctx.msm_pippenger_dev(d_s.data_ptr(), d_p.data_ptr(), nn, o.data_ptr())
        if mode.startswith("both"):
            ctx.stream_wait()
        kept[it] = o
    torch.cuda.synchronize()
    ctx.sync()
    g = bytes(kept.cpu().numpy())
    print("%-28s mismatches %3d / 200" % (mode, sum(g[64 * it:64 * it + 64] != sets[it % 3][2] for it in range(200))))

print("\n-- world-1 step latency, gpu_sharded_msm at n = 5 (host time per call, result left on the device) --")


def legacy_sharded(ctx, d_s, d_p, cnt):
    part = torch.zeros(sv.G1_PARTIAL_BYTES, dtype=torch.uint8, device="cuda")
    out = torch.zeros(64, dtype=torch.uint8, device="cuda")
    legacy_partial(ctx, part, d_s, d_p, cnt)
    torch.cuda.current_stream().synchronize()
    ctx.fold_partials_dev(part.data_ptr(), 1, out.data_ptr())
    return out


for name, fn in (("host syncs (r05)", lambda: legacy_sharded(ctx, ds, dp, n)), ("events (r06)", lambda: D.gpu_sharded_msm(ctx, ds, dp, n))):
    for _ in range(20):
        fn()
    torch.cuda.synchronize()
    ctx.sync()
    t0 = time.perf_counter()
    for _ in range(300):
        r = fn()
    t_issue = (time.perf_counter() - t0) / 300
    torch.cuda.synchronize()
    ctx.sync()
    t_all = (time.perf_counter() - t0) / 300
    assert bytes(r.cpu().numpy()) == want
    print("%-18s issue %.1f us / call, complete %.1f us / call" % (name, 1e6 * t_issue, 1e6 * t_all))
ctx.close()
