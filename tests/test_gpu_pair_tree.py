"""GPU parity of the PAIR LEVEL (csrc/pair_tree.h + k_pair_fwd / k_binv_* / k_pair_bwd in csrc/msm_pippenger.hip): the
batched-affine addition level in front of the bucket accumulation (`buckets[d-1].add_assign(base)`, reference
snark-verifier/src/util/msm.rs:291-296).  It changes HOW the bucket sums are formed, never the group element: every case
here forces it on (SNARKV_PAIR_TREE=2: also for one MSM at a time) and compares with the C oracle bit for bit, and with
the same call with the level off."""
import os

import pytest

import bn254 as O
import coracle as C

pytestmark = pytest.mark.gpu


@pytest.fixture(params=["2", "3"], ids=["two-kernel", "fused"])
def tree_on(monkeypatch, request):
    """2 = the level as its own kernels + a half-length stream; 3 = the FUSED form (k_pairrun_fwd / k_accumulate_pairs: the
    backward pass adds the pair sums straight into the bucket accumulators; runs of 64 / 96 entries, i.e. >= 2^20 points
    or a context with the throughput hint -- other run lengths fall back to form 2)"""
    monkeypatch.setenv("SNARKV_PAIR_TREE", request.param)
    return request.param


def test_golden_cases_with_pair_level(gpu_ctx, golden_msm, tree_on):
    for case in golden_msm:
        s, p, exp = bytes.fromhex(case["scalars"]), bytes.fromhex(case["points"]), bytes.fromhex(case["expected"])
        assert gpu_ctx.msm_pippenger(s, p) == exp, case["name"]


@pytest.mark.parametrize("n", [1, 2, 3, 31, 64, 65, 1000, 4097, (1 << 16) + 3, 1 << 18])
def test_ragged_sizes_vs_c_oracle(gpu_ctx, n, tree_on):
    import snark_verifier_amd as sv

    s, p = C.sample_scalars(300 + n, n), C.sample_points(400 + n, n)
    exp = C.msm_pippenger(s, p, 8)
    assert gpu_ctx.msm_pippenger(s, p) == exp
    hinted = sv.Context(0)
    hinted.set_throughput_hint(True)  # 96-entry runs: the fused form at every size
    assert hinted.msm_pippenger(s, p) == exp
    hinted.close()


def test_exceptional_pairs_inside_buckets(gpu_ctx, tree_on, monkeypatch):
    """Pairs the affine formula cannot add: equal points (doubling), opposite points (the pair cancels to the identity:
    a SKIP entry of the half-length stream), identity bases, and buckets that consist of such pairs only."""
    n = 6000
    base = C.sample_points(0x71, 50)
    pts = [base[64 * (i % 50):64 * (i % 50) + 64] for i in range(n)]  # every point 120 times: equal pairs in most buckets
    for i in range(0, n, 7):
        pts[i] = O.g1_to_bytes(O.g1_neg(O.g1_from_bytes(pts[i])))     # ... and opposite ones
    for i in range(0, n, 97):
        pts[i] = bytes(64)                                            # identity bases contribute nothing
    p = b"".join(pts)
    import snark_verifier_amd as sv

    hinted = sv.Context(0)
    hinted.set_throughput_hint(True)
    for name, sc in (
        ("all_same_scalar", [0x1234567] * n),                 # one bucket per window holds everything
        ("two_values", [5 + (i & 1) for i in range(n)]),
        ("random", None),
        ("r_minus_small", [O.R - 1 - (i % 3) for i in range(n)]),
        ("zeros_and_ones", [i % 2 for i in range(n)]),
    ):
        s = C.sample_scalars(0x72, n) if sc is None else b"".join(O.fe_to_bytes(x) for x in sc)
        exp = C.msm_pippenger(s, p, 8)
        assert gpu_ctx.msm_pippenger(s, p) == exp, name
        assert hinted.msm_pippenger(s, p) == exp, name
        monkeypatch.setenv("SNARKV_PAIR_TREE", "0")
        assert gpu_ctx.msm_pippenger(s, p) == exp, name
        monkeypatch.setenv("SNARKV_PAIR_TREE", tree_on)
    # P and -P with the SAME scalar, adjacent in every bucket: whole buckets cancel
    half = C.sample_points(0x73, 500)
    both = b"".join(half[64 * i:64 * i + 64] + O.g1_to_bytes(O.g1_neg(O.g1_from_bytes(half[64 * i:64 * i + 64]))) for i in range(500))
    sc = C.sample_scalars(0x74, 500)
    s2 = b"".join(sc[32 * i:32 * i + 32] * 2 for i in range(500))
    assert gpu_ctx.msm_pippenger(s2, both) == bytes(64)
    assert hinted.msm_pippenger(s2, both) == bytes(64)
    assert gpu_ctx.msm_pippenger(s2 + O.fe_to_bytes(9), both + O.g1_to_bytes(O.G1_GEN)) == O.g1_to_bytes(O.g1_mul(O.G1_GEN, 9))
    assert hinted.msm_pippenger(s2 + O.fe_to_bytes(9), both + O.g1_to_bytes(O.G1_GEN)) == O.g1_to_bytes(O.g1_mul(O.G1_GEN, 9))
    hinted.close()


def test_window_sizes_and_hint(gpu_ctx, tree_on):
    """different window sizes = different level-2 bin counts (the pad room of a key) and run lengths"""
    import torch

    import snark_verifier_amd as sv

    n = 1 << 15
    s, p = C.sample_scalars(0x81, n), C.sample_points(0x82, n)
    exp = C.msm_pippenger(s, p, 8)
    ds = torch.frombuffer(bytearray(s), dtype=torch.uint8).cuda()
    dp = torch.frombuffer(bytearray(p), dtype=torch.uint8).cuda()
    out = torch.zeros(64, dtype=torch.uint8, device="cuda")
    for c in (4, 8, 10, 13, 16):
        gpu_ctx.msm_pippenger_dev(ds.data_ptr(), dp.data_ptr(), n, out.data_ptr(), c)
        gpu_ctx.sync()
        assert bytes(out.cpu().numpy()) == exp, c
    ctx = sv.Context(0)
    ctx.set_throughput_hint(True)  # 96-entry runs
    assert ctx.msm_pippenger(s, p) == exp
    ctx.close()


def test_2p20_bench_seeds_bit_exact_with_pair_level(gpu_ctx, tree_on):
    """BASELINE config 2 at its size through the pair level (8.4 M affine additions, two inversion levels + the final
    product tree), against the threaded C restatement of util/msm.rs:308-343"""
    import torch

    n = 1 << 20
    ds = torch.empty(32 * n, dtype=torch.uint8, device="cuda")
    dp = torch.empty(64 * n, dtype=torch.uint8, device="cuda")
    torch.cuda.synchronize()
    gpu_ctx.sample_scalars_dev(0x5EED0001, n, ds.data_ptr())
    gpu_ctx.sample_points_dev(0x5EED0002, n, dp.data_ptr())
    gpu_ctx.sync()
    exp = C.msm_pippenger(bytes(ds.cpu().numpy()), bytes(dp.cpu().numpy()), os.cpu_count() or 1)
    out = torch.zeros(64, dtype=torch.uint8, device="cuda")
    gpu_ctx.msm_pippenger_dev(ds.data_ptr(), dp.data_ptr(), n, out.data_ptr(), 0)
    gpu_ctx.sync()
    assert bytes(out.cpu().numpy()) == exp


@pytest.mark.parametrize("mode", ["1", "3"])
def test_batch_and_chunk_pipeline_with_pair_level(gpu_ctx, monkeypatch, mode):
    """SNARKV_PAIR_TREE=1 (3: the fused form): the level follows the throughput hint -- a batch's job contexts and the chunk pipeline's
    worker lanes always carry it -- and the bytes equal the level-off single calls."""
    import torch

    sizes = [(1 << 16) + 5, 70000, 1 << 17, 12345, (1 << 16) + 5]
    ds, dp, exp = [], [], []
    monkeypatch.setenv("SNARKV_PAIR_TREE", "0")
    for i, n in enumerate(sizes):
        s, p = C.sample_scalars(0x90 + i, n), C.sample_points(0xA0 + i, n)
        ds.append(torch.frombuffer(bytearray(s), dtype=torch.uint8).cuda())
        dp.append(torch.frombuffer(bytearray(p), dtype=torch.uint8).cuda())
        exp.append(gpu_ctx.msm_pippenger(s, p))
        assert exp[-1] == C.msm_pippenger(s, p, 8)
    out = torch.zeros(64 * len(sizes), dtype=torch.uint8, device="cuda")
    monkeypatch.setenv("SNARKV_PAIR_TREE", mode)
    gpu_ctx.msm_pippenger_many_dev([t.data_ptr() for t in ds], [t.data_ptr() for t in dp], sizes, out.data_ptr())
    gpu_ctx.sync()
    assert bytes(out.cpu().numpy()) == b"".join(exp)
    # three 2^20-point chunks + a ragged one through the worker lanes (their contexts carry the hint)
    n = 3 * (1 << 20) + 777
    big_s = torch.empty(32 * n, dtype=torch.uint8, device="cuda")
    big_p = torch.empty(64 * n, dtype=torch.uint8, device="cuda")
    o1 = torch.zeros(64, dtype=torch.uint8, device="cuda")
    gpu_ctx.sample_scalars_dev(0xB1, n, big_s.data_ptr())
    gpu_ctx.sample_points_dev(0xB2, n, big_p.data_ptr())
    gpu_ctx.msm_pippenger_dev(big_s.data_ptr(), big_p.data_ptr(), n, o1.data_ptr(), 0)
    gpu_ctx.sync()
    with_level = bytes(o1.cpu().numpy())
    monkeypatch.setenv("SNARKV_PAIR_TREE", "0")
    gpu_ctx.msm_pippenger_dev(big_s.data_ptr(), big_p.data_ptr(), n, o1.data_ptr(), 0)
    gpu_ctx.sync()
    assert with_level == bytes(o1.cpu().numpy()) != bytes(64)
