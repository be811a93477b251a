"""The default window size of the Pippenger (csrc/msm_pippenger.hip `default_window_bits`, exported through
`snarkv_g1_msm_bucket_geometry`; no device needed): chosen by a cost model among the sizes whose TOP window is populated.
A narrow top window piles a whole window's entries into a few level-1 keys, each sorted by one workgroup -- 2^19 points at
the round-1 choice c = 15 spent 1.4 ms there instead of 0.12 (profiles/r02_sweep_window_bits.txt)."""
import snark_verifier_amd as sv

DIGIT_BITS = 128  # GLV half-scalars: 127 magnitude bits + the recoding carry


def test_geometry_is_consistent_and_the_top_window_is_populated():
    for lg in range(0, 27):
        for n in {1 << lg, (1 << lg) + 1, 3 << max(lg - 1, 0)}:
            c, w, b = sv.Context.bucket_geometry(n)
            assert 2 <= c <= 22 and b == 1 << (c - 1) and w == -(-DIGIT_BITS // c), (n, c, w, b)
            top = DIGIT_BITS - 1 - (w - 1) * c
            assert top >= c - 3, (n, c, top)


def test_measured_choices():
    pick = lambda n: sv.Context.bucket_geometry(n)[0]
    assert [pick(1 << k) for k in (12, 14, 15, 16, 17, 18, 19, 20, 21, 22)] == [10, 10, 13, 13, 13, 16, 16, 16, 16, 16]
    # an explicit window size is honoured as given
    assert sv.Context.bucket_geometry(1 << 20, 14)[:2] == (14, 10)
