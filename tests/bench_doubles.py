"""TEST INFRASTRUCTURE for tests/test_bench_dry_run.py: a CPU double of `snark_verifier_amd.Context` with exactly the
entry points bench.py's multi-process path calls.  "Device pointers" are host addresses of CPU tensors; the arithmetic is
the C oracle's (oracle/coracle.py).  Loaded by `bench.py --dry-run-doubles <this file>`; never part of the product."""
import ctypes
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "oracle"))
import coracle as C  # noqa: E402

PB = 144  # G1_PARTIAL_BYTES: here an affine point zero-padded (the fold double adds affine points)


class Context:
    def __init__(self, device):
        self.calls = []

    # ---- sampling (the device sampler IS the oracle's sampler: tests/test_gpu_fullsize.py)
    def sample_scalars_dev(self, seed, n, d_out, first=0):
        ctypes.memmove(d_out, C.sample_scalars(seed, n, first=first), 32 * n)

    def sample_points_dev(self, seed, n, d_out, first=0):
        ctypes.memmove(d_out, C.sample_points(seed, n, first=first), 64 * n)

    def wait_stream(self, stream=None):  # `snarkv_ctx_wait_stream` / `snarkv_stream_wait_ctx`: nothing to order on the CPU
        pass

    def stream_wait(self, stream=None):
        pass

    def sync(self):
        pass

    def set_throughput_hint(self, on):
        pass

    def set_stage_timing(self, on):
        pass

    def get_stage_timing(self):
        return {"total": 1.0, "bucket_accumulate": 0.5, "bucket_combine": 0.1}

    @staticmethod
    def launch_points(n, window_bits=0):
        return n

    # ---- one MSM
    def _msm(self, d_s, d_p, n):
        return C.msm_pippenger(ctypes.string_at(d_s, 32 * n), ctypes.string_at(d_p, 64 * n), 1)

    def msm_pippenger_dev(self, d_s, d_p, n, d_out, window_bits=0):
        ctypes.memmove(d_out, self._msm(d_s, d_p, n), 64)

    def msm_pippenger_partial_dev(self, d_s, d_p, n, d_part, window_bits=0):
        ctypes.memmove(d_part, self._msm(d_s, d_p, n) + bytes(PB - 64), PB)

    def fold_partials_dev(self, d_parts, count, d_out):
        raw = ctypes.string_at(d_parts, PB * count)
        acc = bytes(64)
        for r in range(count):
            acc = C.g1_add(acc, raw[PB * r:PB * r + 64])
        ctypes.memmove(d_out, acc, 64)

    # ---- the batch
    def msm_pippenger_many_dev(self, d_s, d_p, counts, d_out, window_bits=0):
        for i, (s, p, n) in enumerate(zip(d_s, d_p, counts)):
            ctypes.memmove(d_out + 64 * i, self._msm(s, p, n), 64)

    def msm_pippenger_many_partial_dev(self, d_s, d_p, counts, d_parts, window_bits=0):
        self.calls.append(list(counts))
        for i, (s, p, n) in enumerate(zip(d_s, d_p, counts)):
            ctypes.memmove(d_parts + PB * i, self._msm(s, p, n) + bytes(PB - 64), PB)

    def fold_partials_many_dev(self, d_by_job, world, k, d_out):
        for i in range(k):
            self.fold_partials_dev(d_by_job + PB * world * i, world, d_out + 64 * i)
