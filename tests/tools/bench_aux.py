#!/usr/bin/env python3
"""Secondary measurements (dev tool): batched small-MSM accumulation shapes
(BASELINE configs 3/5) and the pairing decider.  Prints JSON lines."""
import json
import os
import sys
import time

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "oracle"))


def main():
    import torch

    import snark_verifier_amd as sv
    import bn254 as O

    ctx = sv.Context(0, stream=torch.cuda.current_stream().cuda_stream)

    def t_ms(fn, reps=5, warm=2):
        for _ in range(warm):
            fn()
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        for _ in range(reps):
            fn()
        torch.cuda.synchronize()
        return (time.perf_counter() - t0) / reps * 1e3

    for nproofs in (64, 1024):
        # per proof a 21-term and a 3-term MSM (Gwc19), then KzgAs: two (m+1)-term MSMs
        offs = [0]
        for _ in range(nproofs):
            offs += [offs[-1] + 21, offs[-1] + 24]
        n1 = offs[-1]
        offs2 = [0, nproofs + 1, 2 * (nproofs + 1)]
        n2 = offs2[-1]
        ds = torch.empty(32 * max(n1, n2), dtype=torch.uint8, device="cuda")
        dp = torch.empty(64 * max(n1, n2), dtype=torch.uint8, device="cuda")
        ctx.sample_scalars_dev(1, max(n1, n2), ds.data_ptr())
        ctx.sample_points_dev(2, max(n1, n2), dp.data_ptr())
        o1 = torch.tensor(offs, dtype=torch.int32, device="cuda")
        o2 = torch.tensor(offs2, dtype=torch.int32, device="cuda")
        out1 = torch.zeros(64 * (len(offs) - 1), dtype=torch.uint8, device="cuda")
        out2 = torch.zeros(128, dtype=torch.uint8, device="cuda")
        torch.cuda.synchronize()

        def run():
            ctx.msm_batched_dev(ds.data_ptr(), dp.data_ptr(), o1.data_ptr(), len(offs) - 1, n1, out1.data_ptr())
            ctx.msm_batched_dev(ds.data_ptr(), dp.data_ptr(), o2.data_ptr(), 2, n2, out2.data_ptr())

        ms = t_ms(run)
        print(json.dumps({"workload": "accumulate %d StandardPlonk+GWC proofs (2x%d small MSMs + KzgAs 2x%d terms)" % (nproofs, nproofs, nproofs + 1),
                          "ms": ms, "proofs_per_s": nproofs / ms * 1e3, "terms": n1 + n2}))

    sec = 0x1234567
    dk = sv.DecidingKey(ctx, O.g1_to_bytes(O.G1_GEN), O.g2_to_bytes(O.G2_GEN), O.g2_to_bytes(O.g2_mul(O.G2_GEN, sec)))
    acc = O.g1_to_bytes(O.g1_mul(O.G1_GEN, sec)) + O.g1_to_bytes(O.G1_GEN)
    for m in (1, 64, 1024):
        da = torch.frombuffer(bytearray(acc * m), dtype=torch.uint8).cuda()
        ok = torch.zeros(m, dtype=torch.uint8, device="cuda")
        torch.cuda.synchronize()
        ms = t_ms(lambda: ctx.decide_batch_dev(dk, da.data_ptr(), m, ok.data_ptr()), reps=3, warm=1)
        assert bool(ok.cpu().all())
        print(json.dumps({"workload": "decide_all over %d accumulators" % m, "ms": ms, "decides_per_s": m / ms * 1e3}))


if __name__ == "__main__":
    main()
