"""Dev tool: one LARGE MSM (2^k points) three ways -- single launch, the chunk-split form of the library
(SNARKV_PIP_SPLIT=1), and 2^20-point chunks spread by hand over `--lanes` independent contexts (partial + fold):
what the chunked form COULD reach.  python tools/bench_large_msm.py 24 [--lanes 4]"""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
import snark_verifier_amd as sv

k = int(sys.argv[1]) if len(sys.argv) > 1 else 24
lanes = int(sys.argv[sys.argv.index("--lanes") + 1]) if "--lanes" in sys.argv else 4
n = 1 << k
ctx = sv.Context(0)
ds = torch.empty(32 * n, dtype=torch.uint8, device="cuda"); dp = torch.empty(64 * n, dtype=torch.uint8, device="cuda")
out = torch.zeros(64, dtype=torch.uint8, device="cuda")
torch.cuda.synchronize()
ctx.sample_scalars_dev(1, n, ds.data_ptr()); ctx.sample_points_dev(2, n, dp.data_ptr()); ctx.sync()


def timed(fn, reps=4):
    fn(); torch.cuda.synchronize()
    t = time.perf_counter()
    for _ in range(reps):
        fn()
    torch.cuda.synchronize()
    return (time.perf_counter() - t) / reps * 1e3


def single():
    ctx.msm_pippenger_dev(ds.data_ptr(), dp.data_ptr(), n, out.data_ptr(), 0); ctx.sync()


os.environ["SNARKV_PIP_SPLIT"] = "0"
t0 = timed(single); ref = bytes(out.cpu().numpy())
os.environ["SNARKV_PIP_SPLIT"] = "2"
t1 = timed(single); assert bytes(out.cpu().numpy()) == ref
os.environ["SNARKV_PIP_SPLIT"] = "0"
cs = [sv.Context(0, ordered=False) for _ in range(lanes)]
chunk = 1 << 20
nch = n // chunk
parts = torch.zeros(nch * sv.G1_PARTIAL_BYTES, dtype=torch.uint8, device="cuda")


def by_hand():
    for c in range(nch):
        cs[c % lanes].msm_pippenger_partial_dev(ds.data_ptr() + 32 * c * chunk, dp.data_ptr() + 64 * c * chunk, chunk,
                                                parts.data_ptr() + c * sv.G1_PARTIAL_BYTES, 0)
    for c in cs:
        c.sync()
    ctx.fold_partials_dev(parts.data_ptr(), nch, out.data_ptr()); ctx.sync()


t2 = timed(by_hand); assert bytes(out.cpu().numpy()) == ref
print("2^%d: single launch %.2f ms (%.3e pts/s) | library split %.2f ms | %d hand lanes %.2f ms (%.3e pts/s)" %
      (k, t0, n / t0 * 1e3, t1, lanes, t2, n / t2 * 1e3))
