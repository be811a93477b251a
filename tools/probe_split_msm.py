"""Probe (GPU box): the succinct-verify MSMs of N proofs (21 + 3 terms each) as ONE bn254_g1_msm_batched call against the same
terms split over T concurrent calls from T host threads -- the question behind pipelining a Keccak aggregation job
(profiles/r06_probe_split_msm.txt: one call of 1 024 proofs 1.08 ms; 2 x 512: 1.71; 4 x 256: 1.82 -- concurrent small
launches share the chip badly, so only a job whose chunks are spread over time (the Poseidon pipeline) gains)."""
import ctypes, os, sys, threading, time
sys.path.insert(0, os.getcwd())
os.environ.setdefault("GPU_MAX_HW_QUEUES", "16")
import torch
import snark_verifier_amd as sv
lib = sv.load_library()
def make(nproofs):
    offs = [0]
    for _ in range(nproofs):
        offs += [offs[-1] + 21, offs[-1] + 24]
    n1 = offs[-1]
    ctx = sv.Context(0)
    ds = torch.empty(32 * n1, dtype=torch.uint8, device="cuda"); dp = torch.empty(64 * n1, dtype=torch.uint8, device="cuda")
    torch.cuda.synchronize()
    ctx.sample_scalars_dev(0x5EED0003, n1, ds.data_ptr()); ctx.sample_points_dev(0x5EED0004, n1, dp.data_ptr()); ctx.sync()
    hs, hp = bytes(ds.cpu().numpy()), bytes(dp.cpu().numpy()); ctx.close()
    return offs, n1, hs, hp
def run(nproofs, T, reps=12):
    offs, n1, hs, hp = make(nproofs)
    o1 = (ctypes.c_uint32 * len(offs))(*offs)
    times = []
    def worker(k, bar, out):
        ps, pp = ctypes.c_void_p(), ctypes.c_void_p()
        assert lib.bn254_host_buffer(0, 32 * n1, ctypes.byref(ps)) == 0 and lib.bn254_host_buffer(1, 64 * n1, ctypes.byref(pp)) == 0
        ctypes.memmove(ps, hs, 32 * n1); ctypes.memmove(pp, hp, 64 * n1)
        ps2, pp2 = ctypes.cast(ps, ctypes.c_char_p), ctypes.cast(pp, ctypes.c_char_p)
        o = ctypes.create_string_buffer(64 * (len(offs) - 1))
        for r in range(reps + 2):
            bar.wait()
            t0 = time.perf_counter()
            assert lib.bn254_g1_msm_batched(ps2, pp2, o1, len(offs) - 1, o) == 0
            out[k].append(time.perf_counter() - t0)
            bar.wait()
    bar = threading.Barrier(T)
    out = [[] for _ in range(T)]
    ts = [threading.Thread(target=worker, args=(k, bar, out)) for k in range(T)]
    [t.start() for t in ts]; [t.join() for t in ts]
    per_round = [max(out[k][r] for k in range(T)) for r in range(2, reps + 2)]
    per_round.sort()
    return per_round[len(per_round) // 2] * 1e3
for nproofs, T in ((1024, 1), (512, 2), (256, 4), (128, 8), (256, 1), (128, 1), (64, 1)):
    print("%4d proofs per call x %d concurrent calls: %.3f ms per round (median, slowest thread)" % (nproofs, T, run(nproofs, T)))
