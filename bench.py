#!/usr/bin/env python3
"""bench.py -- BN254 G1 MSM points/sec at 2^20 on MI355X (BASELINE.json metric).

A "step" is ONE pass of the hot path over one batch of synthetic input: every
rank reduces its own 2^20-point shard (inputs resident in HBM, generated there
by the seeded SplitMix64 sampler of SURVEY.md 8d) with the HIP Pippenger.  The K
timed steps are submitted as ONE batch call (`snarkv_g1_msm_pippenger_many_dev`:
the library pipelines the K MSMs -- sorts on high-priority streams, bucket
accumulations back to back on three streams, one batched tail -- every job on
its own input arrays); for N > 1 every rank computes K projective partials, ONE
all-gather moves K x 144 B per rank over RCCL/xGMI and every rank folds them to
the same K affine points.  value = N * 2^20 * steps / wall time
(max over ranks) -> weak scaling.  `--inflight N` keeps N single-MSM calls in
flight instead (the round-1 way, for comparison); `config.single_msm_latency_ms`
gives the strictly sequential figure.

Launch:  python bench.py [--gpus N --steps K --warmup W]
         python -m torch.distributed.run --nnodes=1 --nproc-per-node N ... bench.py --gpus N ...
Prints ONE JSON line on rank 0.
"""
import argparse
import json
import os
import sys
import time

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)
# hardware queues the HIP runtime spreads this process's streams over (default 4): the package sets the same default on
# import; here too because the runtime reads it at its first call, which torch may make first (see snark-verifier_amd/__init__.py)
os.environ.setdefault("GPU_MAX_HW_QUEUES", "16")

AGG_JOBS_IN_FLIGHT = 16  # aggregation jobs kept in flight (one context each) in the `_pipelined` lines of the second metric
HBM_PEAK_GBPS = 8000.0  # MI355X_MICROARCH.md: 8.0 TB/s spec (6.29 TB/s measured copy ceiling)
BYTES_PER_POINT = 96    # 64 B affine point + 32 B scalar, read once


def _kernel_source_hash():
    import importlib.util

    spec = importlib.util.spec_from_file_location("_srchash", os.path.join(ROOT, "snark-verifier_amd", "_srchash.py"))
    m = importlib.util.module_from_spec(spec)
    spec.loader.exec_module(m)
    return m.kernel_source_hash()


def _profile_record(pattern):
    """The newest profiles/<pattern> JSON whose recorded kernel-source hash equals the tree's, else (None, why)."""
    import glob

    cur = _kernel_source_hash()
    stale = None
    for path in sorted(glob.glob(os.path.join(ROOT, "profiles", pattern)), key=os.path.getmtime, reverse=True):
        try:
            rec = json.load(open(path))
        except Exception:
            continue
        if rec.get("kernel_source_hash") == cur:
            return rec, os.path.relpath(path, ROOT)
        stale = stale or os.path.relpath(path, ROOT)
    return None, ("no record for kernel sources %s (newest: %s, taken on other kernels)" % (cur, stale)) if stale else \
        "no record under profiles/%s" % pattern


def dist_ms(samples):
    """individually timed calls -> the estimator every timing key states: the headline is the MEDIAN"""
    xs = sorted(samples)
    n = len(xs)
    med = xs[n // 2] if n % 2 else 0.5 * (xs[n // 2 - 1] + xs[n // 2])
    return {"estimator": "median", "calls": n, "median_ms": med, "min_ms": xs[0], "mean_ms": sum(xs) / n,
            "p95_ms": xs[min(n - 1, -(-95 * n // 100) - 1)], "max_ms": xs[-1]}


def cpu_baseline(ctx, d_scalars, d_points, sample_log2):
    """Reference algorithm restated in C (oracle/c/bn254_oracle.c <- util/msm.rs:259-343),
    timed on the host cores over a bounded sample of the SAME inputs.
    Not a halo2curves measurement."""
    sys.path.insert(0, os.path.join(ROOT, "oracle"))
    import coracle  # the only use of oracle/ in this file: the reported CPU baseline

    m = 1 << sample_log2
    s = bytes(d_scalars[: 32 * m].cpu().numpy())
    p = bytes(d_points[: 64 * m].cpu().numpy())
    # Threads: rayon sizes its pool by `available_parallelism`, which honours the container's CPU quota (cgroup cpu.max)
    # -- the reference on THIS box would run 16 threads, not 256.  Both counts are timed; `value` is the better one, with
    # its thread count in `cores`.
    ncpu = os.cpu_count() or 1
    quota = _cgroup_cpu_quota()
    counts = [ncpu] + ([max(1, int(quota + 0.999))] if quota and quota + 0.999 < ncpu else [])
    runs = []
    for c in counts:
        t0 = time.perf_counter()
        out = coracle.msm_pippenger(s, p, c)
        runs.append((time.perf_counter() - t0, c))
    dt, cores = min(runs)
    # the reference without its `parallel` feature is single-threaded (msm.rs:308-343): T = 1 on a smaller sample
    m1 = min(m, 1 << 16)
    t1 = time.perf_counter()
    coracle.msm_pippenger(s[: 32 * m1], p[: 64 * m1], 1)
    dt1 = time.perf_counter() - t1
    return {
        "single_thread": {"value": m1 / dt1, "unit": "points/s", "cores": 1,
                          "sample": "first 2^%d points of the same inputs, 1 thread, %.2f s" % (m1.bit_length() - 1, dt1)},
        "value": m / dt,
        "unit": "points/s",
        "cores": cores,
        "kind": "port",
        "sample": "2^%d points of the same seeded inputs, util::msm Pippenger restated in C "
                  "(window ceil(ln n)+2, chunk-per-thread as msm.rs:311-336), %d threads, %.2f s; "
                  "not a halo2curves measurement" % (sample_log2, cores, dt),
        "cgroup_cpu_quota": quota,
        "by_thread_count": {str(c): m / t for t, c in runs},
    }, out, (s, p)


def secondary_metrics(sv, torch, ctxs, cpu=True):
    """BASELINE.json's second metric ("aggregated proofs verified/sec") on shape-faithful
    synthetic work: per proof a 21-term and a 3-term MSM
    (Gwc19::verify), then KzgAs::verify's two (m+1)-term MSMs, then ONE decide.
    All EC work on the device through the same C-ABI entry points the C++ host
    mirror uses; the host-side Fr algebra (microseconds per proof) is not included.
    Reported twice: one aggregation job at a time (latency: three dependent
    latency-bound launches) and with one job in flight per context (throughput)."""
    out = {}
    ctx = ctxs[0]

    def t_ms(fn, reps=5, warm=2):
        for _ in range(warm):
            fn()
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        for _ in range(reps):
            fn()
        torch.cuda.synchronize()
        return (time.perf_counter() - t0) / reps * 1e3

    def t_each(fn, calls):  # every call timed on its own, synchronised: what ONE caller waiting for ONE result sees
        xs = []
        for _ in range(calls):
            torch.cuda.synchronize()
            t0 = time.perf_counter()
            fn()
            torch.cuda.synchronize()
            xs.append((time.perf_counter() - t0) * 1e3)
        return dist_ms(xs)

    # deciding key: any valid (g1, g2, s*g2); the toy secret only matters for accept/reject
    g2 = bytes.fromhex(
        "edf692d95cbdde46ddda5ef7d422436779445c5e66006a42761e1f12efde0018c212f3aeb785e49712e7a9353349aaf1255dfb31b7bf60723a480d9293938e19"
        "aa7dfa6601cce64c7bd3430c69e7d1e38f40cb8d8071ab4aeb6d8cdba55ec8125b9722d1dcdaac55f38eb37033314bbc95330c69ad999eec75f05f58d0890609")
    g1 = (1).to_bytes(32, "little") + (2).to_bytes(32, "little")
    dks = [sv.DecidingKey(c, g1, g2, g2) for c in ctxs]  # s = 1: (P, P) is a valid accumulator
    dk = dks[0]
    for nproofs in (64, 1024):
        offs = [0]
        for _ in range(nproofs):
            offs += [offs[-1] + 21, offs[-1] + 24]
        n1, n2 = offs[-1], 2 * (nproofs + 1)
        ds = torch.empty(32 * max(n1, n2), dtype=torch.uint8, device="cuda")
        dp = torch.empty(64 * max(n1, n2), dtype=torch.uint8, device="cuda")
        ctx.sample_scalars_dev(0x5EED0003, max(n1, n2), ds.data_ptr())
        ctx.sample_points_dev(0x5EED0004, max(n1, n2), dp.data_ptr())
        o1 = torch.tensor(offs, dtype=torch.int32, device="cuda")
        o2 = torch.tensor([0, nproofs + 1, n2], dtype=torch.int32, device="cuda")
        out1 = [torch.zeros(64 * (len(offs) - 1), dtype=torch.uint8, device="cuda") for _ in ctxs]
        acc = [torch.zeros(128, dtype=torch.uint8, device="cuda") for _ in ctxs]
        ok = [torch.zeros(1, dtype=torch.uint8, device="cuda") for _ in ctxs]
        torch.cuda.synchronize()

        def job(k):
            c = ctxs[k]
            c.msm_batched_dev(ds.data_ptr(), dp.data_ptr(), o1.data_ptr(), len(offs) - 1, n1, out1[k].data_ptr())
            c.msm_batched_dev(ds.data_ptr(), dp.data_ptr(), o2.data_ptr(), 2, n2, acc[k].data_ptr())
            c.decide_batch_dev(dks[k], acc[k].data_ptr(), 1, ok[k].data_ptr())

        ms = t_ms(lambda: job(0), reps=10)
        out["aggregate_%d_proofs" % nproofs] = {"ms": ms, "proofs_per_s": nproofs / ms * 1e3,
                                                 "msm_terms": n1 + n2, "includes_decide": True,
                                                 "jobs_in_flight": 1, "estimator": "mean", "calls": 10,
                                                 "timing": "mean of 10 calls enqueued back to back on one stream, ONE job at a time (the named config)",
                                                 "per_call": t_each(lambda: job(0), 25),
                                                 "roofline": aggregate_roofline(nproofs, ms)}
        if len(ctxs) > 1:
            for c in ctxs:  # other launches are in flight next to each context's: the library's throughput hint (the
                c.set_throughput_hint(True)  # joint fixed-window form above 16 384 terms: less work, a longer chain)

            def wave():
                for k in range(len(ctxs)):
                    job(k)
            ms = min(t_ms(wave, reps=4, warm=1) for _ in range(3)) / len(ctxs)  # best of three timed regions of 4 x 16 jobs
            same = all(bytes(a.cpu().numpy()) == bytes(acc[0].cpu().numpy()) for a in acc[1:])
            out["aggregate_%d_proofs_pipelined" % nproofs] = {
                "ms_per_job": ms, "proofs_per_s": nproofs / ms * 1e3, "msm_terms": n1 + n2,
                "includes_decide": True, "jobs_in_flight": len(ctxs), "hw_queues": os.environ.get("GPU_MAX_HW_QUEUES"),
                "results_identical": same, "timing": "best of 3 timed regions of 4 x 16 jobs", "best_of": 3, "estimator": "min", "calls": 3,
                "NOT_the_named_config": "16 jobs in flight on 16 contexts / hardware queues: a throughput figure",
                "roofline": aggregate_roofline(nproofs, ms)}
            for c in ctxs:
                c.set_throughput_hint(False)
        # ... and AGG_JOBS_IN_FLIGHT jobs MERGED into one set of launches (what snarkv_host_aggregate_many does with the
        # jobs of one call): one segmented launch for every proof's MSMs, one for the 2 J KzgAs MSMs, one decide_batch(J)
        J = AGG_JOBS_IN_FLIGHT
        offs_m = [0]
        for _ in range(nproofs * J):
            offs_m += [offs_m[-1] + 21, offs_m[-1] + 24]
        offs2_m = [k * (nproofs + 1) for k in range(2 * J + 1)]
        n1m, n2m = offs_m[-1], offs2_m[-1]
        dsm = torch.empty(32 * max(n1m, n2m), dtype=torch.uint8, device="cuda")
        dpm = torch.empty(64 * max(n1m, n2m), dtype=torch.uint8, device="cuda")
        ctx.sample_scalars_dev(0x5EED0003, max(n1m, n2m), dsm.data_ptr())
        ctx.sample_points_dev(0x5EED0004, max(n1m, n2m), dpm.data_ptr())
        o1m = torch.tensor(offs_m, dtype=torch.int32, device="cuda")
        o2m = torch.tensor(offs2_m, dtype=torch.int32, device="cuda")
        out1m = torch.zeros(64 * (len(offs_m) - 1), dtype=torch.uint8, device="cuda")
        accm = torch.zeros(128 * J, dtype=torch.uint8, device="cuda")
        okm = torch.zeros(J, dtype=torch.uint8, device="cuda")
        torch.cuda.synchronize()

        def merged():
            ctx.msm_batched_dev(dsm.data_ptr(), dpm.data_ptr(), o1m.data_ptr(), len(offs_m) - 1, n1m, out1m.data_ptr())
            ctx.msm_batched_dev(dsm.data_ptr(), dpm.data_ptr(), o2m.data_ptr(), 2 * J, n2m, accm.data_ptr())
            ctx.decide_batch_dev(dk, accm.data_ptr(), J, okm.data_ptr())

        ms = min(t_ms(merged, reps=2, warm=1) for _ in range(3)) / J
        out["aggregate_%d_proofs_merged" % nproofs] = {
            "ms_per_job": ms, "proofs_per_s": nproofs / ms * 1e3, "msm_terms": (n1m + n2m) // J, "includes_decide": True,
            "jobs_merged": J, "launch_sets": 1, "timing": "best of 3 timed regions of 2 merged calls", "best_of": 3, "estimator": "min", "calls": 3,
            "NOT_the_named_config": "16 jobs merged into one set of launches: a throughput figure",
            "roofline": aggregate_roofline(nproofs, ms)}
        del dsm, dpm, out1m
        if cpu:  # 64 proofs ~0.15 s, 1 024 proofs ~2.5 s of one host thread
            out["aggregate_%d_proofs" % nproofs]["cpu_baseline"] = cpu_baseline_aggregate(ds, dp, offs, n1, nproofs, n2)
    one = torch.frombuffer(bytearray((g1 + g1) * 1024), dtype=torch.uint8).cuda()
    oks = torch.zeros(1024, dtype=torch.uint8, device="cuda")
    torch.cuda.synchronize()
    for m in (1, 1024):
        ms = t_ms(lambda: ctx.decide_batch_dev(dk, one.data_ptr(), m, oks.data_ptr()), reps=10, warm=2)
        out["decide_all_%d" % m] = {"ms": ms, "decides_per_s": m / ms * 1e3, "all_accept": bool(oks[:m].cpu().all()),
                                    "estimator": "mean", "calls": 10, "timing": "mean of 10 calls enqueued back to back",
                                    "per_call": t_each(lambda: ctx.decide_batch_dev(dk, one.data_ptr(), m, oks.data_ptr()), 25)}
    if cpu:
        out["decide_all_1"]["cpu_baseline"], out["decide_all_1024"]["cpu_baseline"] = cpu_baseline_decide(g2, g1 + g1)
    for d in dks:
        d.close()
    # `IpaAs::decide` (pcs/ipa/decider.rs:47-55) at k = 20: the other consumer of the 2^20-point MSM --
    # committing key resident on the device, h_coeffs built by a kernel, 20 scalars in / 64 bytes out
    k = 20
    gpts = torch.empty(64 << k, dtype=torch.uint8, device="cuda")
    torch.cuda.synchronize()
    ctx.sample_points_dev(0x5EED0005, 1 << k, gpts.data_ptr())
    ctx.sync()
    ipa_dk = sv.IpaDecidingKey(ctx, bytes(gpts.cpu().numpy()))
    xi = b"".join((0x1234567 * (i + 3)).to_bytes(32, "little") for i in range(k))
    u = bytes(64)
    ms = t_ms(lambda: ctx.ipa_decide_batch(ipa_dk, xi, u), reps=5, warm=2)
    out["ipa_decide_k20"] = {"ms": ms, "terms": 1 << k, "bytes_in": 32 * k + 64, "estimator": "mean", "calls": 5,
                             "note": "h_coeffs kernel + one 2^20-term Pippenger over the resident committing key"}
    ipa_dk.close()
    del gpts
    return out


def host_resident_metrics(sv, torch, ctx, d_scalars_k, d_points_k, n, steps, slot_results, nsets):
    """The drop-in rate a caller holding HOST slices sees (`util::msm::multi_scalar_multiplication(&[Fr], &[G1Affine])`,
    util/msm.rs:308): K jobs whose scalars / points lie in pinned host memory, through snarkv_g1_msm_pippenger_many (uploads
    on a copy stream, a job's kernels wait for its own upload only), against the link's own rate measured with the same
    buffers -- and the same batch with SNARKV_FLAG_MONTGOMERY (halo2curves' in-memory form: no per-element conversion on
    the host at all).  Never `value`: that is the HBM-resident rate."""
    import ctypes

    sets = min(4, nsets)  # 96 MiB of pinned memory each
    hs = ctx.host_buffer(0, 32 * n * sets)
    hp = ctx.host_buffer(1, 64 * n * sets)
    def to_pinned(dst_addr, dev_tensor, nbytes):
        a = dev_tensor.cpu().numpy()  # (kept alive across the copy)
        ctypes.memmove(dst_addr, a.ctypes.data, nbytes)

    for k in range(sets):
        to_pinned(ctypes.addressof(hs) + 32 * n * k, d_scalars_k[k], 32 * n)
        to_pinned(ctypes.addressof(hp) + 64 * n * k, d_points_k[k], 64 * n)
    ps = [ctypes.addressof(hs) + 32 * n * (i % sets) for i in range(steps)]
    pp = [ctypes.addressof(hp) + 64 * n * (i % sets) for i in range(steps)]
    res = ctx.msm_pippenger_many_host(ps, pp, [n] * steps)  # initialisation: staging buffers, job contexts
    ok = all(res[i] == slot_results[i % sets] for i in range(steps) if i % sets < len(slot_results))
    best = None
    for _ in range(3):
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        res = ctx.msm_pippenger_many_host(ps, pp, [n] * steps)
        dt = time.perf_counter() - t0
        best = dt if best is None else min(best, dt)
    # the link alone: the same uploads with no kernel behind them
    dst = torch.empty(96 * n, dtype=torch.uint8, device="cuda")
    torch.cuda.synchronize()
    src_s = torch.frombuffer(hs, dtype=torch.uint8)
    src_p = torch.frombuffer(hp, dtype=torch.uint8)
    link = None
    for _ in range(3):
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        for i in range(steps):
            k = i % sets
            dst[: 32 * n].copy_(src_s[32 * n * k: 32 * n * (k + 1)], non_blocking=True)
            dst[32 * n:].copy_(src_p[64 * n * k: 64 * n * (k + 1)], non_blocking=True)
        torch.cuda.synchronize()
        d = time.perf_counter() - t0
        link = d if link is None else min(link, d)
    out = {
        "value": n * steps / best, "unit": "points/s", "ms_per_msm": best / steps * 1e3, "jobs": steps, "best_of": 3,
        "entry_point": "snarkv_g1_msm_pippenger_many (host pointers; pinned memory from snarkv_ctx_host_buffer)",
        "results_match_the_device_resident_batch": ok,
        "pcie_bound": {"points_per_s": n * steps / link, "gb_per_s": 96.0 * n * steps / link / 1e9, "ms_per_msm": link / steps * 1e3,
                       "how": "the same %d x 96 MiB uploads from the same pinned buffers with no kernel behind them (torch copy_, "
                              "non_blocking), best of 3" % steps},
        "fraction_of_pcie_bound": link / best,
    }
    # the same batch in halo2curves' in-memory form: inputs sampled by a context whose default is SNARKV_FLAG_MONTGOMERY
    mctx = sv.Context(torch.cuda.current_device())
    mctx.set_flags(sv.SNARKV_FLAG_MONTGOMERY)
    ms_t, mp_t = torch.empty(32 * n, dtype=torch.uint8, device="cuda"), torch.empty(64 * n, dtype=torch.uint8, device="cuda")
    torch.cuda.synchronize()
    mctx.sample_scalars_dev(0x5EED0001, n, ms_t.data_ptr())  # input set 0 again, the other encoding
    mctx.sample_points_dev(0x5EED0002, n, mp_t.data_ptr())
    mctx.sync()
    hs2, hp2 = mctx.host_buffer(0, 32 * n), mctx.host_buffer(1, 64 * n)
    to_pinned(ctypes.addressof(hs2), ms_t, 32 * n)
    to_pinned(ctypes.addressof(hp2), mp_t, 64 * n)
    a_s, a_p = [ctypes.addressof(hs2)] * steps, [ctypes.addressof(hp2)] * steps
    r = mctx.msm_pippenger_many_host(a_s, a_p, [n] * steps)
    bestm = None
    for _ in range(3):
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        r = mctx.msm_pippenger_many_host(a_s, a_p, [n] * steps)
        d = time.perf_counter() - t0
        bestm = d if bestm is None else min(bestm, d)
    # decode the Montgomery result on the host (one point) and compare with the canonical one
    P = 21888242871839275222246405745257275088696311157297823662689037894645226208583
    dec = b"".join((int.from_bytes(r[0][j:j + 32], "little") * pow(1 << 256, -1, P) % P).to_bytes(32, "little") for j in (0, 32))
    out["in_memory_form"] = {"value": n * steps / bestm, "unit": "points/s", "ms_per_msm": bestm / steps * 1e3,
                             "flags": "SNARKV_FLAG_MONTGOMERY", "decoded_result_matches": dec == slot_results[0],
                             "note": "scalars and points as halo2curves holds them in memory (a * 2^256 in 4 x u64), result likewise; "
                                     "every job reads the one pinned input set"}
    mctx.close()
    return out


def aggregate_roofline(nproofs, ms):
    """SURVEY.md 8(d) for the second metric: one proof = 24 (scalar, point) pairs x 96 B = 2 304 B, + the KzgAs step's
    2 x (m + 1) pairs x 96 B, + 128 B for the decided accumulator -- algorithmic bytes of the whole job over its
    duration, against the HBM peak.  The job is three dependent latency-bound launches (small MSMs, KzgAs MSMs, one
    pairing), so the fraction is ~1e-5: the figure is reported as prescribed, the kernel shares are in
    profiles/r03_rocprofv3_kernel_stats_aggregate.csv (dominant: k_decide)."""
    alg = 2304 * nproofs + 2 * (nproofs + 1) * 96 + 128
    ach = alg / (ms * 1e-3) / 1e9
    return {"bound": "hbm", "algorithmic_bytes": alg, "achieved": ach, "peak": HBM_PEAK_GBPS, "unit": "GB/s",
            "frac": ach / HBM_PEAK_GBPS, "dominant_kernel": _aggregate_dominant_kernel(nproofs)}


def _aggregate_dominant_kernel(nproofs):
    """the longest kernel of the aggregation job by total duration, read from the committed rocprofv3 summary"""
    import csv
    import glob

    paths = sorted(glob.glob(os.path.join(ROOT, "profiles", "r*_rocprofv3_kernel_stats_aggregate_%d.csv" % nproofs)))
    if not paths:
        return None
    try:
        rows = list(csv.DictReader(open(paths[-1])))
        top = max(rows, key=lambda r: float(r["TotalDurationUs"]))
        return {"name": top["Name"].split("(")[0], "share": float(top["Percentage"]) / 100.0,
                "avg_us": float(top["AverageUs"]), "from_profile": os.path.relpath(paths[-1], ROOT),
                "note": "read from the committed rocprofv3 summary, not measured by this run"}
    except Exception as e:  # a malformed record must not break the bench line
        return {"error": str(e)}


E2E_CALLS = 25  # individually timed calls behind every end_to_end_* key


def _cgroup_cpu_quota():
    """CPUs' worth of time this container may use per scheduling period (cgroup v2 cpu.max), or None if unlimited"""
    try:
        q, per = open("/sys/fs/cgroup/cpu.max").read().split()[:2]
        return None if q == "max" else float(q) / float(per)
    except (OSError, ValueError):
        return None


def _cgroup_throttled():
    try:
        for ln in open("/sys/fs/cgroup/cpu.stat"):
            if ln.startswith("nr_throttled"):
                return int(ln.split()[1])
    except OSError:
        pass
    return None


def _e2e_entry(run, calls, nproofs, extra):
    """`run()` -> (ok, accumulator(s), timings dict): `calls` calls, each timed by the library's own wall clock around the whole
    job; the headline `ms` is the MEDIAN call, with its phase split; min / p95 / max beside it.
    ONE job at a time means: a job arrives when the machine is idle.  Under a container CPU quota (cgroup cpu.max: this
    pool's boxes grant 16 CPUs' worth of time per 100 ms to a 256-thread host) jobs issued back to back run into the quota --
    a 1 024-proof job is ~0.2 CPU-seconds on 64 host threads -- and the kernel then freezes the whole process for tens of
    milliseconds: the calls are therefore PACED (idle time after each call until its CPU time fits the quota), and the line
    says so: `cpu_ms_per_call`, `cgroup_cpu_quota`, `paced_idle_ms_per_call`, `cgroup_throttle_events`, and the rate the
    quota sustains (`quota_bound_ms_per_call` = cpu_ms_per_call / quota)."""
    for _ in range(3):  # warm: pools, scratch, pinned buffers, the first touch of this fixture's buffers (profiles/r05_host_outliers.txt)
        run()
    quota = _cgroup_cpu_quota()
    thr0 = _cgroup_throttled()
    recs, cpu_ms, idle_ms = [], 0.0, 0.0
    for _ in range(calls):
        c0, w0 = time.process_time(), time.perf_counter()
        r = run()
        c, w = (time.process_time() - c0) * 1e3, (time.perf_counter() - w0) * 1e3
        cpu_ms += c
        if not r[0]:
            return {"error": "verifier rejected"}
        recs.append(r)
        if quota:  # idle until this call's CPU time has been paid for at 80 % of the quota
            idle = c / (0.8 * quota) - w
            if idle > 0.05:
                time.sleep(idle * 1e-3)
                idle_ms += idle
    thr1 = _cgroup_throttled()
    recs.sort(key=lambda r: r[-1]["total"])
    med = recs[len(recs) // 2]
    tm = med[-1]
    d = dist_ms([r[-1]["total"] for r in recs])
    out = {"ms": d["median_ms"], "proofs_per_s": nproofs / d["median_ms"] * 1e3}
    out.update(d)
    out.update({"phases_of": "the median call", "ms_read_proofs": tm["read_proofs"], "ms_fr_algebra_host": tm["fr_algebra"],
                "ms_msm_device_incl_h2d": tm["msm_device"], "ms_kzg_accumulate": tm["accumulate"], "ms_decide": tm["decide"],
                "accepted": True, "cpu_ms_per_call": cpu_ms / calls})
    if tm["read_proofs"] + tm["fr_algebra"] + tm["msm_device"] + tm["accumulate"] + tm["decide"] > 1.05 * tm["total"]:
        out["phases_overlap"] = ("pipelined job (host/aggregation.hpp aggregate_pipelined): read / algebra / msm are the helper "
                                 "threads' busy times and run UNDER ms_kzg_accumulate; ms is wall time")
    if quota:
        out.update({"cgroup_cpu_quota": quota, "paced_idle_ms_per_call": idle_ms / calls,
                    "cgroup_throttle_events": None if thr0 is None or thr1 is None else thr1 - thr0,
                    "quota_bound_ms_per_call": cpu_ms / calls / quota})
    out.update(extra)
    return out, med


def end_to_end_metrics():
    """The C3/C5 workloads as REAL INPUT BYTES through the host mirror's verifier API
    (snark-verifier_amd/host/plonk.hpp, transcript.hpp, pcs.hpp): proof bytes ->
    transcript -> expression evaluation -> ONE segmented MSM launch for all
    proofs -> KzgAs accumulation (on a transcript of the proofs' family) -> one pairing decide.  Inputs: the committed
    fixtures tests/golden/bench_plonk_*.bin (StandardPlonk-shaped proofs forged under
    a toy SRS by tests/golden/gen_bench_proofs.py; data only, no oracle code runs here).
    Host work (transcript, Fr algebra) is INCLUDED in these timings.  Every key: E2E_CALLS individually timed calls,
    `ms` = the median, with min / p95 / max."""
    from snark_verifier_amd import host_api as H

    threads = max(1, min(64, os.cpu_count() or 1))
    out = {}
    G = os.path.join(ROOT, "tests", "golden")
    # steady state of a verifier service: the host pool's workers awake, the device contexts of the boundary warm (the
    # phases before this one ran Python threads and left the pool asleep: profiles/r05_host_outliers.txt)
    p0 = os.path.join(G, "bench_plonk_gwc19_evm_64.bin")
    if os.path.exists(p0):
        fx0 = H.read_fixture(p0)
        hp0, hdk0 = H.Protocol(fx0["protocol"]), H.DecidingKey(fx0["dk"])
        for _ in range(20):
            H.aggregate(hp0, hdk0, fx0["instances"], fx0["proofs"], fx0["n"], H.MOS_GWC19, 0, threads)
        hp0.close()
        hdk0.close()
    tnames = {0: "", 1: "_poseidon_transcript_host_hashed", 2: "_poseidon_transcript_device_hashed", 3: "_poseidon_transcript_auto"}
    read_key = {0: "read on the host (Keccak)", 1: "read and hashed on the host", 2: "read incl. the device hashing",
                3: "read, hashed on the host or the device by batch size and host width (auto)"}
    for tkind, tname in ((0, "evm"), (1, "poseidon"), (2, "poseidon"), (3, "poseidon")):
        path = os.path.join(G, "bench_plonk_gwc19_%s_64.bin" % tname)
        if not os.path.exists(path):
            continue
        fx = H.read_fixture(path)
        n = fx["n"]
        hp, hdk = H.Protocol(fx["protocol"]), H.DecidingKey(fx["dk"])
        for rep in (1, 16):
            if rep == 16 and tkind == 1:
                continue  # (1 024 proofs hashed on the host IS what the auto route takes now: the key below it)
            r = _e2e_entry(lambda: H.aggregate(hp, hdk, fx["instances"] * rep, fx["proofs"] * rep, n * rep, H.MOS_GWC19, tkind,
                                               threads, timings=True), E2E_CALLS, n * rep,
                           {"host_threads": threads, "proofs_are": read_key[tkind],
                            "accumulation_transcript": "Keccak" if tkind == 0 else "Poseidon (host; one sponge over 4 m elements)",
                            "input": os.path.basename(path) + (" x%d" % rep if rep > 1 else "")})
            if isinstance(r, dict):
                return r
            e, med = r
            e["matches_fixture_accumulator"] = (med[1] == fx["expected_acc"]) if rep == 1 else None
            out["end_to_end_aggregate_%d_proofs" % (n * rep) + tnames[tkind]] = e
        hp.close()
        hdk.close()
    # 16 jobs of 64 proofs in ONE call (`snarkv_host_aggregate_many`: a service batching its requests): three device
    # launches whatever the number of jobs, so small jobs share them -- compare end_to_end_aggregate_64_proofs x 16
    path = os.path.join(G, "bench_plonk_gwc19_evm_64.bin")
    if os.path.exists(path):
        fx = H.read_fixture(path)
        hp, hdk = H.Protocol(fx["protocol"]), H.DecidingKey(fx["dk"])
        J = 16
        r = _e2e_entry(lambda: H.aggregate_many(hp, hdk, fx["instances"] * J, fx["proofs"] * J, [fx["n"]] * J, H.MOS_GWC19, 0,
                                                threads, timings=True), E2E_CALLS, fx["n"] * J,
                       {"jobs": J, "host_threads": threads, "input": os.path.basename(path) + " x16 jobs"})
        if not isinstance(r, dict):
            e, med = r
            e["ms_per_job"] = e["ms"] / J
            e["matches_fixture_accumulator"] = all(a == fx["expected_acc"] for a in med[1])
            out["end_to_end_aggregate_16_jobs_of_64_proofs_one_call"] = e
        hp.close()
        hdk.close()
    # ... and 16 APPLICATION threads each aggregating 64 proofs end to end through the host C API at the same time (round 5:
    # no process-wide device lock any more -- the mirror's device calls go through the device library's context pool; the
    # host passes still take turns on the one host pool)
    if os.path.exists(path):
        import threading

        fx = H.read_fixture(path)
        hp, hdk = H.Protocol(fx["protocol"]), H.DecidingKey(fx["dk"])
        T, reps, bad = 16, 6, []

        def app(k, reps_):
            for _ in range(reps_):
                ok, acc = H.aggregate(hp, hdk, fx["instances"], fx["proofs"], fx["n"], H.MOS_GWC19, 0, threads)
                if not ok or acc != fx["expected_acc"]:
                    bad.append(k)

        def wave(reps_):
            ts = [threading.Thread(target=app, args=(k, reps_)) for k in range(T)]
            t0 = time.perf_counter()
            for t in ts:
                t.start()
            for t in ts:
                t.join()
            return (time.perf_counter() - t0) * 1e3

        wave(2)
        ms = min(wave(reps) for _ in range(3)) / (T * reps)
        out["end_to_end_aggregate_64_proofs_16_app_threads"] = {
            "ms_per_job": ms, "proofs_per_s": fx["n"] / ms * 1e3, "app_threads": T, "jobs_per_thread": reps, "host_threads_per_call": threads,
            "host_threads_note": "the calls' host passes take turns on the one host pool: wide and short beats narrow and long "
                                 "(measured per job: 1 thread per call 0.40 ms, 4: 0.61, 16: 0.34, 64: 0.34)",
            "estimator": "min", "calls": 3, "timing": "wall time of 16 threads x %d calls of snarkv_host_aggregate, best of 3 regions" % reps,
            "matches_fixture_accumulator": not bad, "input": os.path.basename(path),
            "NOT_the_named_config": "16 jobs in flight through the host C API: a throughput figure; one call alone: end_to_end_aggregate_64_proofs"}
        hp.close()
        hdk.close()
    # config 5 on its own inputs: 1 024 DISTINCT proofs.  The reference's example hashes with POSEIDON, the snarks and the
    # accumulation proof alike (evm-verifier-with-accumulator.rs:361,375): that fixture is the named figure; the Keccak one
    # (what an outer EVM flow would feed) beside it.
    for kind, tkind, key in (("poseidon", 3, "end_to_end_aggregate_1024_distinct_proofs_poseidon"),
                             ("evm", 0, "end_to_end_aggregate_1024_distinct_proofs")):
        path = os.path.join(G, "bench_plonk_gwc19_%s_1024.bin" % kind)
        if not os.path.exists(path):
            continue
        fx = H.read_fixture(path)
        hp, hdk = H.Protocol(fx["protocol"]), H.DecidingKey(fx["dk"])
        r = _e2e_entry(lambda: H.aggregate(hp, hdk, fx["instances"], fx["proofs"], fx["n"], H.MOS_GWC19, tkind, threads, timings=True),
                       E2E_CALLS, fx["n"],
                       {"host_threads": threads, "input": os.path.basename(path),
                        "transcripts": "Poseidon for the proofs (auto route: hashed on the host threads, pipelined with the MSMs) AND "
                                       "for the accumulation proof (one host sponge over 4 096 elements = 1 025 dependent "
                                       "permutations, absorbing chunk by chunk UNDER the reading and the MSMs)"
                                       if kind == "poseidon" else "Keccak for the proofs and the accumulation proof"})
        if not isinstance(r, dict):
            e, med = r
            e["matches_fixture_accumulator"] = med[1] == fx["expected_acc"]
            out[key] = e
        hp.close()
        hdk.close()
    # the SDK's default scheme: SHPLONK = KzgAs<Bn256, Bdfg21> on Poseidon transcripts (snark-verifier-sdk/src/lib.rs:41):
    # 64 proofs (20 + 1-term MSMs, two batch inversions per proof on the host), and the same batch x16
    path = os.path.join(G, "bench_plonk_bdfg21_poseidon_64.bin")
    if os.path.exists(path):
        fx = H.read_fixture(path)
        hp, hdk = H.Protocol(fx["protocol"]), H.DecidingKey(fx["dk"])
        for rep in (1, 16):
            r = _e2e_entry(lambda: H.aggregate(hp, hdk, fx["instances"] * rep, fx["proofs"] * rep, fx["n"] * rep, H.MOS_BDFG21, 3,
                                               threads, timings=True), E2E_CALLS, fx["n"] * rep,
                           {"host_threads": threads, "scheme": "Bdfg21 (SHPLONK), Poseidon transcripts (auto route)",
                            "input": os.path.basename(path) + (" x%d" % rep if rep > 1 else "")})
            if not isinstance(r, dict):
                e, med = r
                e["matches_fixture_accumulator"] = (med[1] == fx["expected_acc"]) if rep == 1 else None
                out["end_to_end_aggregate_%d_proofs_bdfg21_poseidon" % (fx["n"] * rep)] = e
        hp.close()
        hdk.close()
    return out


def context_free_metrics(sv, nproofs=64, T=16):
    """The trait boundary's own throughput: `EcPointLoader::multi_scalar_multiplication` has no `&self`
    (loader.rs:108), so a Rust caller binds the context-free `bn254_*` entry points.  T host threads (a rayon pool's
    workers) each run `nproofs`-proof aggregation jobs through them -- bn254_g1_msm_batched x2 + bn254_kzg_dk_decide_batch, inputs
    packed in the calling thread's pinned buffers (bn254_host_buffer), results on the host after every call -- and the
    library hands every call a context of its pool.  Compare `aggregate_<n>_proofs_pipelined` (16 explicit contexts,
    device-resident inputs, asynchronous calls)."""
    import ctypes
    import threading

    lib = sv.load_library()
    g2 = bytes.fromhex(
        "edf692d95cbdde46ddda5ef7d422436779445c5e66006a42761e1f12efde0018c212f3aeb785e49712e7a9353349aaf1255dfb31b7bf60723a480d9293938e19"
        "aa7dfa6601cce64c7bd3430c69e7d1e38f40cb8d8071ab4aeb6d8cdba55ec8125b9722d1dcdaac55f38eb37033314bbc95330c69ad999eec75f05f58d0890609")
    g1 = (1).to_bytes(32, "little") + (2).to_bytes(32, "little")
    dk = ctypes.c_void_p()
    if lib.bn254_kzg_dk_create(g1, g2, g2, ctypes.byref(dk)) != 0:
        return {"error": sv.last_error()}
    offs = [0]
    for _ in range(nproofs):
        offs += [offs[-1] + 21, offs[-1] + 24]
    n1, n2 = offs[-1], 2 * (nproofs + 1)
    # one input set for all threads (sampled on the device once; every thread copies it into ITS pinned buffers)
    import torch
    ctx = sv.Context(0)
    ds = torch.empty(32 * n1, dtype=torch.uint8, device="cuda")
    dp = torch.empty(64 * n1, dtype=torch.uint8, device="cuda")
    torch.cuda.synchronize()
    ctx.sample_scalars_dev(0x5EED0003, n1, ds.data_ptr())
    ctx.sample_points_dev(0x5EED0004, n1, dp.data_ptr())
    ctx.sync()
    hs, hp = bytes(ds.cpu().numpy()), bytes(dp.cpu().numpy())
    ctx.close()
    o1 = (ctypes.c_uint32 * len(offs))(*offs)
    o2 = (ctypes.c_uint32 * 3)(0, nproofs + 1, n2)
    errs, firsts = [], [None] * T

    def worker(k, reps):
        ps, pp = ctypes.c_void_p(), ctypes.c_void_p()
        if lib.bn254_host_buffer(0, 32 * n1, ctypes.byref(ps)) or lib.bn254_host_buffer(1, 64 * n1, ctypes.byref(pp)):
            errs.append("host_buffer")
            return
        ctypes.memmove(ps, hs, 32 * n1)
        ctypes.memmove(pp, hp, 64 * n1)
        ps, pp = ctypes.cast(ps, ctypes.c_char_p), ctypes.cast(pp, ctypes.c_char_p)  # (the argtypes of the byte-pointer parameters)
        out1, acc, ok = ctypes.create_string_buffer(64 * (len(offs) - 1)), ctypes.create_string_buffer(128), ctypes.create_string_buffer(1)
        for _ in range(reps):
            rc = lib.bn254_g1_msm_batched(ps, pp, o1, len(offs) - 1, out1) or lib.bn254_g1_msm_batched(ps, pp, o2, 2, acc)
            rc = rc or min(0, lib.bn254_kzg_dk_decide_batch(dk, g1 + g1, 1, ok))
            if rc:
                errs.append(rc)
        firsts[k] = out1.raw + acc.raw

    def wave(nthreads, reps):
        ts = [threading.Thread(target=worker, args=(k, reps)) for k in range(nthreads)]
        t0 = time.perf_counter()
        for t in ts:
            t.start()
        for t in ts:
            t.join()
        return (time.perf_counter() - t0) * 1e3

    wave(T, 2)  # pool contexts, scratch, pinned buffers
    reps = 8
    ms = min(wave(T, reps) for _ in range(3)) / (T * reps)
    one = min(wave(1, reps) for _ in range(3)) / reps
    created, cap = ctypes.c_int(0), ctypes.c_int(0)
    lib.bn254_default_contexts(ctypes.byref(created), ctypes.byref(cap))
    lib.snarkv_dk_destroy(dk)
    return {"ms_per_job": ms, "proofs_per_s": nproofs / ms * 1e3, "host_threads": T, "jobs_per_thread": reps,
            "estimator": "min", "calls": 3, "timing": "wall time of %d threads x %d jobs, best of 3 regions" % (T, reps),
            "ms_per_job_one_thread": one, "default_contexts_created": created.value, "default_contexts_cap": cap.value,
            "results_identical": len(set(firsts)) == 1 and not errs, "errors": errs[:4],
            "entry_points": "bn254_g1_msm_batched x2 + bn254_kzg_dk_decide_batch per job, host pointers (the calling thread's "
                            "pinned buffers), synchronous",
            "NOT_the_named_config": "%d jobs in flight through the context-free boundary: a throughput figure" % T}


def cpu_baseline_decide(g2, acc):
    """CPU leg of `decide` / `decide_all(1024)` (pcs/kzg/decider.rs:70-93): the pairing
    decider restated in C (oracle/c/bn254_pairing.inc: G2Prepared x2 per call as the reference, multi Miller loop, final
    exponentiation), one thread as the reference runs it, and spread over all host threads (beyond-reference)."""
    sys.path.insert(0, os.path.join(ROOT, "oracle"))
    import coracle  # cpu_baseline leg only

    reps = 16
    t0 = time.perf_counter()
    for _ in range(reps):
        ok1 = coracle.kzg_decide(g2, g2, acc)
    dt1 = (time.perf_counter() - t0) / reps
    cores = os.cpu_count() or 1
    m = 1024
    t0 = time.perf_counter()
    allok, _ = coracle.kzg_decide_all(g2, g2, acc * m, cores)
    dtm = time.perf_counter() - t0
    one = {"value": 1.0 / dt1, "unit": "decides/s", "cores": 1, "kind": "port", "accepted": bool(ok1),
           "sample": "one KzgAs::decide (2 x G2Prepared::from + 2-pair Miller loop + final exponentiation) restated in C, "
                     "1 thread, mean of %d calls = %.3f ms; not a halo2curves measurement" % (reps, dt1 * 1e3)}
    many = {"value": m / dtm, "unit": "decides/s", "cores": cores, "kind": "port", "accepted": bool(allok),
            "single_thread": {"value": 1.0 / dt1, "unit": "decides/s", "cores": 1,
                              "sample": "decide_all is a loop of decide in the reference (decider.rs:84-93): 1 024 x %.3f ms = %.2f s on one thread"
                                        % (dt1 * 1e3, m * dt1)},
            "sample": "decide_all over 1 024 accumulators, the same C restatement spread over %d host threads (beyond-reference "
                      "threading), %.3f s; not a halo2curves measurement" % (cores, dtm)}
    return one, many


def cpu_baseline_aggregate(ds, dp, offs, n1, nproofs, n2):
    """CPU leg of the aggregate metric: the reference's naive
    NativeLoader loop (native.rs:61-71) restated in C, ONE thread (the reference has no threading on this
    path), followed by ONE pairing decide of the C restatement (decider.rs:70-82) -- the same job the device runs."""
    sys.path.insert(0, os.path.join(ROOT, "oracle"))
    import coracle  # cpu_baseline leg only

    s = bytes(ds[: 32 * max(n1, n2)].cpu().numpy())
    p = bytes(dp[: 64 * max(n1, n2)].cpu().numpy())
    t0 = time.perf_counter()
    coracle.msm_batched(s[: 32 * n1], p[: 64 * n1], offs)
    acc = coracle.msm_batched(s[: 32 * n2], p[: 64 * n2], [0, nproofs + 1, n2])
    g2 = bytes.fromhex(
        "edf692d95cbdde46ddda5ef7d422436779445c5e66006a42761e1f12efde0018c212f3aeb785e49712e7a9353349aaf1255dfb31b7bf60723a480d9293938e19"
        "aa7dfa6601cce64c7bd3430c69e7d1e38f40cb8d8071ab4aeb6d8cdba55ec8125b9722d1dcdaac55f38eb37033314bbc95330c69ad999eec75f05f58d0890609")
    t1 = time.perf_counter()
    coracle.kzg_decide(g2, g2, acc)
    dt_dec = time.perf_counter() - t1
    dt = time.perf_counter() - t0
    return {"value": nproofs / dt, "unit": "proofs/s", "cores": 1, "kind": "port", "includes_decide": True,
            "sample": "the %d-proof job's %d MSM terms, naive double-and-add loop of native.rs:61-71 restated in C, then one "
                      "pairing decide (%.1f ms), 1 thread, %.2f s in all; not a halo2curves measurement" % (nproofs, n1 + n2, dt_dec * 1e3, dt)}


# ---- what goes to stdout: ONE compact contract line, the rest to a file ----------------------------------------------
# The driver parses the LAST stdout line and keeps only a few KB of tail (round 5: a 26.5 KB line was cut mid-way and the
# round went unmeasured).  So the last line carries the contract keys, the roofline, the CPU baseline and the named
# configs -- under 4 KB, asserted -- and everything else (stage tables, secondary legs, notes, launch attempts) goes to
# `details` (gpurun_out/bench_details.json unless SNARKV_BENCH_DETAILS says otherwise), which the line names.
LINE_LIMIT = 4096


def _details_path():
    p = os.environ.get("SNARKV_BENCH_DETAILS") or os.path.join(ROOT, "gpurun_out", "bench_details.json")
    os.makedirs(os.path.dirname(os.path.abspath(p)), exist_ok=True)
    return p


def _pick(d, keys):
    return {k: d[k] for k in keys if isinstance(d, dict) and k in d}


def compact_line(line, details):
    cfg = line.get("config", {})
    c = _pick(line, ("metric", "value", "unit", "n_gpus", "steps", "warmup", "ms_per_step", "higher_is_better", "scaling", "vs_baseline"))
    c["dtype"] = "i32x9 (254-bit Montgomery Fq as 9 x 29-bit limbs on the integer VALU)"
    c["data"] = line.get("data")
    cc = _pick(cfg, ("workload", "points_per_gpu", "points_per_kernel_launch", "window_bits", "msms_in_flight", "single_msm_latency_ms",
                     "single_msm_points_per_s", "rccl_ranks_seen", "data_plane_ranks_seen", "ranks_share_devices",
                     "result_matches_one_gpu_recompute", "FALLBACK"))
    cc["timed_region"] = ("K steps = ONE batch call of K MSMs, inputs and the K affine results resident in HBM; the D2H of the "
                          "64-byte results is outside the timed region" if cc.get("msms_in_flight") == line.get("steps") else
                          "single-MSM calls kept in flight; inputs and results in HBM, D2H of the 64-byte results outside the timed region")
    if cfg.get("transport"):
        cc["transport"] = {"kind": cfg["transport"].get("kind", "")[:60], "fallback_reason": (cfg["transport"].get("fallback_reason") or None)
                           and str(cfg["transport"]["fallback_reason"])[:160]}
    if cfg.get("launch"):
        at = cfg["launch"].get("attempts", [])
        cc["launch"] = {"self_launched": True, "attempts": len(at), "how": at[-1]["how"][:80] if at else None}
    c["config"] = cc
    if "roofline" in line:
        c["roofline"] = _pick(line["roofline"], ("bound", "kernel", "achieved", "peak", "unit", "frac", "traffic", "kernel_ms",
                                                 "kernel_us_profile", "frac_from_profile"))
    if "cpu_baseline" in line:
        cb = _pick(line["cpu_baseline"], ("value", "unit", "cores", "kind", "sample", "gpu_matches_on_sample", "sample_is_the_whole_workload", "cgroup_cpu_quota"))
        cb["sample"] = str(cb.get("sample", ""))[:260]
        c["cpu_baseline"] = cb
    if isinstance(line.get("config4_strong"), dict):
        c["config4_strong"] = _pick(line["config4_strong"], ("value", "unit", "n_gpus", "points_per_gpu", "ms_per_step",
                                                             "matches_one_gpu_single_call", "skipped"))
    mg = line.get("single_process_mgpu")
    if isinstance(mg, dict):
        c["single_process_mgpu"] = dict(_pick(mg, ("ranks", "job0_matches_one_gpu_recompute", "transports_agree")),
                                        **{k: (mg[k].get("value") if "value" in mg[k] else "error") for k in ("rccl", "peer_copy")
                                           if isinstance(mg.get(k), dict)})
        if "error" in mg:
            c["single_process_mgpu"]["error"] = str(mg["error"])[:120]
    if "named_configs" in line:
        nc = line["named_configs"]
        c["named_configs"] = {k: v for k, v in nc.items() if not isinstance(v, dict)}
        for k, v in nc.items():
            if isinstance(v, dict):  # the pipelined / merged legs: proofs per s, flat under a prefix that says what they are
                c["named_configs"].update({"jobs_in_flight." + kk: vv for kk, vv in v.items()})
    c["details"] = os.path.relpath(details, ROOT) if os.path.abspath(details).startswith(ROOT + os.sep) else details
    return c


def emit(line):
    """the full record to the details file, the compact contract line as the LAST line of stdout"""
    path = _details_path()
    # the host this ran on: logical CPUs, and what the container may use of them (cgroup cpu.max: every host-threaded figure
    # of the record -- CPU baselines, end-to-end jobs, application-thread legs -- lives inside that budget)
    line.setdefault("host", {"logical_cpus": os.cpu_count(), "cgroup_cpu_quota_cpus": _cgroup_cpu_quota()})
    with open(path, "w") as f:
        json.dump(line, f, indent=1)
    c = compact_line(line, path)
    text = json.dumps(c)
    # the line must NEVER outgrow the driver's log tail (it is ~2.8 KB): should a future key push it over, the optional
    # parts go first, the contract keys last -- dropping is recorded in the line, failing is not an option here
    dropped = []
    for k in ("single_process_mgpu", "config4_strong", "named_configs", "cpu_baseline.sample", "config.timed_region", "config.launch",
              "config.transport"):
        if len(text) < LINE_LIMIT:
            break
        top, _, sub = k.partition(".")
        if sub:
            if isinstance(c.get(top), dict) and sub in c[top]:
                c[top][sub] = "see details"
                dropped.append(k)
        elif top in c:
            c[top] = "see details"
            dropped.append(k)
        if dropped:
            c["dropped_for_size"] = dropped
        text = json.dumps(c)
    if len(text) >= LINE_LIMIT:
        sys.stderr.write("bench.py: the contract line is %d bytes (limit %d)\n" % (len(text), LINE_LIMIT))
    _flush_c_stdio()  # RCCL prints a version banner through C stdio: out now, so that the JSON line is the LAST line
    print(text, flush=True)


def _flush_c_stdio():
    """what C libraries of this process have buffered for stdout / stderr (RCCL's banner) goes out NOW"""
    import ctypes

    try:
        ctypes.CDLL(None).fflush(None)
    except Exception:  # noqa: BLE001 -- cosmetic
        pass


def _free_port():
    import socket

    with socket.socket() as so:
        so.bind(("127.0.0.1", 0))
        return so.getsockname()[1]


def _last_json_line(text):
    for ln in reversed(text.splitlines()):
        if ln.startswith("{"):
            try:
                return json.loads(ln)
            except Exception:
                continue
    return None


def self_launch(args, argv):
    """`python bench.py --gpus N` without a launcher: start the N ranks under torch.distributed.run ourselves and relay
    rank 0's ONE line, with a ladder so that a lease on a multi-GPU node never ends without a figure:
      1. one process per GPU, RCCL over xGMI (each rank probes the communicator; a failure falls back in-process to 2.)
      2. one process per GPU, host-staged gloo exchange (a fresh launch, should 1. have died or hung)
      3. ONE process driving all GPUs through snarkv_mgpu_* with peer copies (no collective library, no launcher)
    Every attempt is recorded in config.launch of the line that is printed."""
    import subprocess

    limit = float(os.environ.get("SNARKV_BENCH_LAUNCH_TIMEOUT", "1500"))
    attempts = []
    base = [a for a in argv]
    for label, extra in (("torch.distributed.run, transport auto (RCCL, in-process fallback to gloo)", []),
                         ("torch.distributed.run, transport gloo (host-staged)", ["--transport", "gloo"])):
        cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", str(args.gpus),
               "--master-addr", "127.0.0.1", "--master-port", str(_free_port()), os.path.abspath(__file__)] + base + extra
        t0 = time.perf_counter()
        try:
            r = subprocess.run(cmd, stdout=subprocess.PIPE, stderr=subprocess.PIPE, text=True, timeout=limit, cwd=ROOT)
            rc, out, err = r.returncode, r.stdout, r.stderr
        except subprocess.TimeoutExpired as e:
            rc, out, err = "timeout after %.0f s" % limit, (e.stdout or b"").decode(errors="replace") if isinstance(e.stdout, bytes) else (e.stdout or ""), \
                (e.stderr or b"").decode(errors="replace") if isinstance(e.stderr, bytes) else (e.stderr or "")
        line = _last_json_line(out) if rc == 0 else None
        attempts.append({"how": label, "rc": rc, "seconds": round(time.perf_counter() - t0, 1),
                         "stderr_tail": None if line else err[-1500:]})
        if line:
            # the child printed the compact line and wrote the full record: the attempts go into both
            full = None
            try:
                with open(os.path.join(ROOT, line["details"]) if not os.path.isabs(line["details"]) else line["details"]) as f:
                    full = json.load(f)
            except Exception:  # noqa: BLE001 -- the record is a convenience; the line is the contract
                full = line
            full.setdefault("config", {})["launch"] = {"self_launched": True, "attempts": attempts}
            emit(full)
            return 0
        sys.stderr.write("bench.py: attempt failed (%s): rc %s\n%s\n" % (label, rc, err[-3000:]))
        if args.dry_run_doubles:
            break  # (the dry run is gloo either way)
    if not args.dry_run_doubles:
        leg = run_mgpu_leg_subprocess(args.gpus, args.steps, args.log2n, args.window_bits, limit)
        best = max((leg.get(k) for k in ("peer_copy", "rccl") if isinstance(leg.get(k), dict) and "value" in leg[k]),
                   key=lambda d: d["value"], default=None)
        attempts.append({"how": "single process, snarkv_mgpu_* (no launcher)", "rc": 0 if best else "failed"})
        if best:
            n = 1 << args.log2n
            line = {"metric": "BN254 G1 MSM points/sec at 2^%d" % args.log2n, "value": best["value"], "unit": "points/s",
                    "n_gpus": args.gpus, "steps": args.steps, "warmup": args.warmup, "ms_per_step": best["ms_per_step"],
                    "higher_is_better": True, "scaling": "weak", "vs_baseline": None,
                    "dtype": "i32x9 (254-bit Montgomery Fq as 9 x 29-bit signed lazy limbs on the integer VALU)", "data": "synthetic",
                    "config": {"workload": "BN254 G1 Pippenger MSM, 2^%d random points/scalars per GPU, inputs resident in HBM, "
                                           "affine result (configs[1])" % args.log2n, "points_per_gpu": n,
                               "parallelism": "point-sharded x%d in ONE process (snarkv_g1_msm_pippenger_many_mgpu_dev), transport %s"
                                              % (args.gpus, best["transport"]),
                               "FALLBACK": "both one-process-per-GPU launches failed; this is the single-process leg",
                               "launch": {"self_launched": True, "attempts": attempts}},
                    "single_process_mgpu": leg}
            emit(line)
            return 0
    sys.stderr.write("bench.py: every launch attempt failed: %s\n" % json.dumps(attempts))
    return 1


def open_data_plane(torch, dist, want, local_rank, world):
    """The group the 144-byte partials travel on.  `auto` / `rccl`: an RCCL ("nccl") group next to the gloo default
    group, opened and PROBED here (an all-reduce of ones, on a helper thread with a deadline: communicator set-up is where a
    misconfigured node fails or hangs); the ranks then agree over gloo -- if ANY rank failed, ALL use the host-staged gloo
    all-gather, and the reason goes into the line.  Returns {"kind": "rccl" | "gloo", "group": ..., "why": ..., ...}."""
    import datetime
    import threading

    from snark_verifier_amd.distributed import set_data_group

    info = {"kind": "gloo", "group": None, "why": None, "requested": want}
    if want == "gloo":
        info["why"] = "requested"
        return info
    res = {"ok": False, "err": None, "group": None}

    def probe():
        try:
            g = dist.new_group(backend="nccl", timeout=datetime.timedelta(seconds=180))
            res["group"] = g
            t = torch.ones(1, dtype=torch.int32, device="cuda")
            dist.all_reduce(t, group=g)
            torch.cuda.synchronize()
            res["ok"] = int(t.item()) == world
            if not res["ok"]:
                res["err"] = "sum of ones over the RCCL group = %d, world = %d" % (int(t.item()), world)
        except Exception as e:  # noqa: BLE001 -- whatever RCCL raises, the ranks fall back together
            res["err"] = "%s: %s" % (type(e).__name__, str(e)[:400])

    def run():
        try:
            with torch.cuda.device(local_rank):
                probe()
        except Exception as e:  # noqa: BLE001
            res["ok"], res["err"] = False, res["err"] or "%s: %s" % (type(e).__name__, str(e)[:400])

    th = threading.Thread(target=run, daemon=True)
    t0 = time.perf_counter()
    th.start()
    th.join(float(os.environ.get("SNARKV_BENCH_RCCL_PROBE_TIMEOUT", "240")))
    if th.is_alive():
        res["ok"], res["err"] = False, "RCCL probe still running after %.0f s (abandoned)" % (time.perf_counter() - t0)
        info["probe_hung"] = True  # main() leaves through os._exit: the stuck thread must not block interpreter shutdown
    flag = torch.tensor([1 if res["ok"] else 0], dtype=torch.int32)
    dist.all_reduce(flag, op=dist.ReduceOp.MIN)  # gloo: every rank learns whether EVERY rank has RCCL
    errs = [None] * world
    dist.all_gather_object(errs, res["err"])
    info["probe_seconds"] = round(time.perf_counter() - t0, 2)
    if int(flag.item()) == 1:
        info.update(kind="rccl", group=res["group"])
        set_data_group(res["group"])
        return info
    info["why"] = "RCCL unavailable on rank(s) %s: %s" % ([r for r, e in enumerate(errs) if e], next((e for e in errs if e), "?"))
    if want == "rccl":
        raise SystemExit("--transport rccl: " + info["why"])
    return info


def run_mgpu_leg_subprocess(gpus, steps, log2n, window_bits, limit=900.0):
    """the single-process leg in a process of its own (no torchrun environment, its own deadline): a hang or a crash of
    RCCL's single-process communicator there cannot take the main line with it"""
    import subprocess

    env = {k: v for k, v in os.environ.items()
           if k not in ("RANK", "LOCAL_RANK", "WORLD_SIZE", "LOCAL_WORLD_SIZE", "GROUP_RANK", "ROLE_RANK", "MASTER_ADDR", "MASTER_PORT",
                        "TORCHELASTIC_RUN_ID", "TORCHELASTIC_RESTART_COUNT", "TORCHELASTIC_MAX_RESTARTS", "GROUP_WORLD_SIZE",
                        "ROLE_WORLD_SIZE", "ROLE_NAME", "TORCHELASTIC_USE_AGENT_STORE", "TORCH_NCCL_ASYNC_ERROR_HANDLING",
                        "TORCHELASTIC_ERROR_FILE")}
    cmd = [sys.executable, os.path.abspath(__file__), "--mgpu-leg", "--gpus", str(gpus), "--steps", str(steps), "--log2n", str(log2n),
           "--window-bits", str(window_bits)]
    try:
        r = subprocess.run(cmd, stdout=subprocess.PIPE, stderr=subprocess.PIPE, text=True, timeout=limit, cwd=ROOT, env=env)
    except subprocess.TimeoutExpired:
        return {"error": "single-process leg timed out after %.0f s" % limit}
    d = _last_json_line(r.stdout)
    if r.returncode != 0 or d is None:
        return {"error": "single-process leg failed: rc %s" % r.returncode, "stderr_tail": r.stderr[-1500:]}
    return d


def mgpu_leg(gpus, steps, log2n, window_bits):
    """The path a Rust / C caller reaches WITHOUT a launcher (include/snarkv_amd.h "multi-GPU in ONE process"): the same
    K-job batch as the main line -- every job 2^log2n points per GPU, resident on its GPU -- through
    snarkv_g1_msm_pippenger_many_mgpu_dev over devices 0..gpus-1, once per transport: one grouped RCCL all-gather of
    K x 144 B per rank (`ncclCommInitAll`), and plain peer copies.  A transport that fails is reported with the
    library's error text and the other one still runs."""
    import torch

    import snark_verifier_amd as sv

    n = 1 << log2n
    ndev = max(1, torch.cuda.device_count())
    devices = [g % ndev for g in range(gpus)]  # (fewer GPUs than ranks: a test box; ranks then share devices and RCCL refuses)
    out = {"ranks": gpus, "devices": devices, "jobs": steps, "points_per_gpu_per_job": n,
           "entry_point": "snarkv_g1_msm_pippenger_many_mgpu_dev", "estimator": "min", "calls": 3}
    try:
        mg = sv.MultiGpu(devices)
    except Exception as e:  # noqa: BLE001
        return dict(out, error="snarkv_mgpu_create: %s" % e)
    en, un, fl = mg.peer_access()
    out["peer_access"] = {"enabled": en, "unavailable": un, "failed": fl,
                          "last_error": sv.last_error() if (un or fl) else None}
    nsets = min(steps, 8)
    keep, ds, dp = [], [], []
    for g in range(gpus):
        c = mg.rank_context(g)
        rs, rp = [], []
        with torch.cuda.device(devices[g]):
            for k in range(nsets):
                s = torch.empty(32 * n, dtype=torch.uint8, device="cuda:%d" % devices[g])
                p = torch.empty(64 * n, dtype=torch.uint8, device="cuda:%d" % devices[g])
                torch.cuda.synchronize()
                first = (k * gpus + g) * n  # job k over all ranks = points [k G n, (k + 1) G n) of the seeded streams
                c.sample_scalars_dev(0x5EED0001, n, s.data_ptr(), first=first)
                c.sample_points_dev(0x5EED0002, n, p.data_ptr(), first=first)
                c.sync()
                keep += [s, p]
                rs.append(s.data_ptr()), rp.append(p.data_ptr())
        ds.append([rs[i % nsets] for i in range(steps)])
        dp.append([rp[i % nsets] for i in range(steps)])
    cn = [[n] * steps for _ in range(gpus)]
    results = {}
    for name, tr in (("rccl", sv.MultiGpu.RCCL), ("peer_copy", sv.MultiGpu.PEER_COPY)):
        try:
            mg.set_transport(tr)
            res = mg.msm_pippenger_many_dev(ds, dp, cn, window_bits)  # initialisation: scratch of every rank's job contexts
            best = None
            for _ in range(3):
                t0 = time.perf_counter()
                res = mg.msm_pippenger_many_dev(ds, dp, cn, window_bits)
                dt = time.perf_counter() - t0
                best = dt if best is None else min(best, dt)
            results[name] = res
            out[name] = {"value": gpus * n * steps / best, "unit": "points/s", "ms_per_step": best / steps * 1e3,
                         "transport": name, "timing": "wall time of the whole call (results on the host), min of 3 calls"}
        except Exception as e:  # noqa: BLE001
            out[name] = {"error": str(e)[:600], "snarkv_last_error": sv.last_error()}
    if len(results) == 2:
        out["transports_agree"] = results["rccl"] == results["peer_copy"]
    if results:
        r = next(iter(results.values()))
        out["results_distinct_per_input_set"] = len(set(r)) == nsets
        out["result_job0"] = r[0].hex()
        # job 0 again on ONE GPU: rank 0 samples all G n points of it and reduces them alone
        c0 = mg.rank_context(0)
        with torch.cuda.device(0):
            s = torch.empty(32 * n * gpus, dtype=torch.uint8, device="cuda:0")
            p = torch.empty(64 * n * gpus, dtype=torch.uint8, device="cuda:0")
            o = torch.zeros(64, dtype=torch.uint8, device="cuda:0")
            torch.cuda.synchronize()
            c0.sample_scalars_dev(0x5EED0001, n * gpus, s.data_ptr(), first=0)
            c0.sample_points_dev(0x5EED0002, n * gpus, p.data_ptr(), first=0)
            c0.msm_pippenger_dev(s.data_ptr(), p.data_ptr(), n * gpus, o.data_ptr(), 0)
            c0.sync()
            out["job0_matches_one_gpu_recompute"] = bytes(o.cpu().numpy()) == r[0]
    mg.close()
    return out


def strong_leg(torch, dist, sv, ctx, stream, world, rank, total_log2n, steps, window_bits, dry, dev, dev_sync):
    """BASELINE config 4 in the same run as the weak line: 2^total_log2n points IN TOTAL, sharded evenly over the ranks,
    `steps` such MSMs as one batch (one all-gather of steps x 144 B per rank).  Rank 0 then reduces MSM 0's whole input
    alone on its GPU and compares: the N-GPU result must be the 1-GPU result."""
    from snark_verifier_amd.distributed import gpu_sharded_msm_batch

    total = 1 << total_log2n
    if total % world:
        return {"skipped": "the world size does not divide 2^%d" % total_log2n}
    n = total // world
    nsets = min(steps, 3)
    d_s = [torch.empty(32 * n, dtype=torch.uint8, device=dev) for _ in range(nsets)]
    d_p = [torch.empty(64 * n, dtype=torch.uint8, device=dev) for _ in range(nsets)]
    dev_sync()
    for k in range(nsets):  # MSM k = points [k total, (k + 1) total) of the seeded streams; this rank holds its n of them
        ctx.sample_scalars_dev(0x5EED0011, n, d_s[k].data_ptr(), first=k * total + rank * n)
        ctx.sample_points_dev(0x5EED0012, n, d_p[k].data_ptr(), first=k * total + rank * n)
    ctx.sync()

    def run():
        return gpu_sharded_msm_batch(ctx, [d_s[i % nsets] for i in range(steps)], [d_p[i % nsets] for i in range(steps)],
                                     [n] * steps, window_bits, stream=stream)

    def barrier():
        if world > 1:
            dist.barrier()
        dev_sync()

    res = run()  # initialisation (scratch) + warm-up
    barrier()
    res = run()
    barrier()
    t0 = time.perf_counter()
    res = run()
    barrier()
    dt = time.perf_counter() - t0
    if world > 1:
        t = torch.tensor([dt], dtype=torch.float64)
        dist.all_reduce(t, op=dist.ReduceOp.MAX)
        dt = float(t.item())
    got = bytes(res.cpu().numpy())
    out = {"metric": "BN254 G1 MSM points/sec at 2^%d IN TOTAL (BASELINE configs[3])" % total_log2n, "scaling": "strong",
           "value": total * steps / dt, "unit": "points/s", "n_gpus": world, "points_per_gpu": n, "steps": steps,
           "ms_per_step": dt / steps * 1e3, "estimator": "one timed region of %d MSMs after one warm-up region, max over ranks" % steps,
           "result": got[:64].hex()}
    if rank == 0 and not dry:
        if world > 1:  # the whole of MSM 0 on this one GPU
            s = torch.empty(32 * total, dtype=torch.uint8, device=dev)
            p = torch.empty(64 * total, dtype=torch.uint8, device=dev)
            dev_sync()
            ctx.sample_scalars_dev(0x5EED0011, total, s.data_ptr(), first=0)
            ctx.sample_points_dev(0x5EED0012, total, p.data_ptr(), first=0)
        else:
            s, p = d_s[0], d_p[0]
        o = torch.zeros(64, dtype=torch.uint8, device=dev)
        dev_sync()
        ctx.msm_pippenger_dev(s.data_ptr(), p.data_ptr(), total, o.data_ptr(), window_bits)
        ctx.sync()
        out["matches_one_gpu_single_call"] = bytes(o.cpu().numpy()) == got[:64]
    return out


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=20)
    ap.add_argument("--warmup", type=int, default=5)
    ap.add_argument("--log2n", type=int, default=20, help="points per GPU = 2^log2n")
    ap.add_argument("--window-bits", type=int, default=0)
    ap.add_argument("--cpu-sample-log2", type=int, default=0,
                    help="CPU baseline sample = the first 2^k points of the same inputs (default 0: the WHOLE workload, so the "
                         "GPU result is compared with the CPU restatement on every point -- at 2^24 through the chunk pipeline)")
    ap.add_argument("--total-log2n", type=int, default=0,
                    help="STRONG scaling: 2^k points in total, split evenly over the ranks (BASELINE config 4: --gpus 8 "
                         "--total-log2n 24); overrides --log2n")
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--no-secondary", action="store_true")
    ap.add_argument("--no-host-resident", action="store_true", help="skip the host-resident (PCIe-inclusive) batch measurement")
    ap.add_argument("--force-dist", action="store_true", help="take the multi-GPU code path even at world size 1 (testing)")
    ap.add_argument("--dry-run-doubles", default="",
                    help="TEST HOOK (tests/test_bench_dry_run.py): path of a module providing CPU doubles of the device context; "
                         "the multi-process control flow of this file (rendezvous, shards, batch call, all-gather, fold, max over "
                         "ranks, the JSON line and its labels) then runs over gloo on CPU tensors.  Not a measurement: the line says so.")
    ap.add_argument("--inflight", type=int, default=0,
                    help="0 (default): the K timed steps are ONE batch call (snarkv_g1_msm_pippenger_many_dev: the library "
                         "pipelines them).  N >= 1: N independent single-MSM calls kept in flight instead (one context + HIP "
                         "stream each; 1 = strictly sequential) -- the round-1 / early round-2 way, kept for comparison")
    ap.add_argument("--transport", choices=("auto", "rccl", "gloo"), default=None,
                    help="N > 1: how the 144-byte partials travel between the ranks.  auto (default): RCCL (torch's \"nccl\" "
                         "backend) over xGMI, probed at start-up (the dry run defaults to gloo); if any rank cannot open it, ALL ranks fall back to a "
                         "host-staged gloo all-gather, and the line says so.  rccl: no fallback.  gloo: host-staged only")
    ap.add_argument("--no-strong", action="store_true", help="skip the config-4 leg (2^24 points IN TOTAL over the ranks)")
    ap.add_argument("--strong-total-log2n", type=int, default=24)
    ap.add_argument("--no-mgpu-leg", action="store_true", help="skip the single-process snarkv_mgpu_* leg")
    ap.add_argument("--mgpu-leg", action="store_true",
                    help="INTERNAL: run only the single-process multi-GPU leg over devices 0..gpus-1 and print its JSON")
    args = ap.parse_args()

    if args.mgpu_leg:
        res = mgpu_leg(args.gpus, args.steps, args.log2n, args.window_bits)
        _flush_c_stdio()
        print(json.dumps(res), flush=True)
        return
    if "WORLD_SIZE" not in os.environ and args.gpus > 1:
        # a plain `python bench.py --gpus N`: launch the N ranks ourselves and relay rank 0's line
        raise SystemExit(self_launch(args, sys.argv[1:]))

    import contextlib

    import torch
    import torch.distributed as dist

    import snark_verifier_amd as sv

    world = int(os.environ.get("WORLD_SIZE", "1"))
    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    if args.gpus != world and rank == 0:
        sys.stderr.write("bench.py: --gpus %d but WORLD_SIZE=%d: the launcher's world size is what runs\n" % (args.gpus, world))
    dry = bool(args.dry_run_doubles)
    if dry:  # CPU doubles of the device context: the control flow of the N > 1 path without a GPU (see --dry-run-doubles)
        import importlib.util

        spec = importlib.util.spec_from_file_location("_bench_doubles", args.dry_run_doubles)
        doubles = importlib.util.module_from_spec(spec)
        spec.loader.exec_module(doubles)
        args.no_cpu_baseline = args.no_secondary = True
        dev = "cpu"
        make_stream = lambda: None  # noqa: E731
        make_ctx = lambda st: doubles.Context(local_rank)  # noqa: E731
        dev_sync = lambda: None  # noqa: E731
        on_stream = lambda st: contextlib.nullcontext()  # noqa: E731
        launch_points = doubles.Context.launch_points
        shared_devices = False
    else:
        # a box with fewer GPUs than ranks (the 1-GPU test box running the N-rank path: tests/test_gpu_bench_ranks.py): the
        # ranks share devices round robin -- RCCL refuses that, so the probe fails on every rank and the run goes on
        # host-staged; the line says `ranks_share_devices`
        ndev = torch.cuda.device_count()
        dev_index = local_rank % max(1, ndev)
        shared_devices = world > ndev
        torch.cuda.set_device(dev_index)
        dev = "cuda"
        make_stream = torch.cuda.Stream
        make_ctx = lambda st: sv.Context(dev_index, stream=st.cuda_stream)  # noqa: E731
        dev_sync = torch.cuda.synchronize
        on_stream = torch.cuda.stream
        launch_points = sv.Context.launch_points
    use_dist = world > 1 or args.force_dist
    if dry and not use_dist:
        raise SystemExit("--dry-run-doubles exercises the multi-process path: launch with torch.distributed.run or add --force-dist")
    transport = {"kind": None}
    if use_dist:
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        os.environ.setdefault("MASTER_PORT", "29511")
        os.environ.setdefault("RANK", "0")
        os.environ.setdefault("WORLD_SIZE", "1")
        # control plane (barriers, max-over-ranks, the fallback agreement): gloo, which needs nothing from the GPUs;
        # data plane (the partials): RCCL, opened and probed behind it -- see open_data_plane
        import datetime

        # a probe that hangs is abandoned by open_data_plane; RCCL's watchdog must not take the process down meanwhile
        os.environ.setdefault("TORCH_NCCL_ASYNC_ERROR_HANDLING", "0")
        # ONE node: RCCL's bootstrap (the out-of-band exchange that sets the communicator up; the data goes over xGMI) needs
        # no network interface beyond loopback and no InfiniBand probe -- the container's hostname may not resolve and its
        # other interfaces are none of this job's business.  An explicit setting of the caller wins.
        os.environ.setdefault("NCCL_SOCKET_IFNAME", "lo")
        os.environ.setdefault("NCCL_IB_DISABLE", "1")
        dist.init_process_group("gloo", timeout=datetime.timedelta(minutes=30))
        transport = open_data_plane(torch, dist, args.transport or ("gloo" if dry else "auto"), local_rank if dry else dev_index, world)

    strong = args.total_log2n > 0
    if strong:
        if (1 << args.total_log2n) % world:
            raise SystemExit("--total-log2n: the world size must divide 2^k")
        n = (1 << args.total_log2n) // world
        args.log2n = n.bit_length() - 1
    else:
        n = 1 << args.log2n
    # One context per in-flight MSM, each on its own HIP stream: the latency-bound
    # tail of one MSM (bucket reduce, 2^(cw) doubling chains, to_affine: a few
    # wavefronts) overlaps the VALU-bound bucket accumulation of the next.
    batch = args.inflight <= 0  # the K steps as one batch call
    inflight = 4 if batch else max(1, args.inflight)
    # explicit side streams only: a context given the NULL stream handle (torch's legacy default
    # stream) would create its own stream, invisible to the stream ordering torch.distributed relies on
    # The second metric keeps AGG_JOBS_IN_FLIGHT aggregation jobs in flight on contexts of their own.  Their streams are
    # created FIRST: the runtime hands its hardware queues to streams in creation order, and 16 streams created after
    # others can end up sharing queues (tools/aggregate_inflight.py --dummy-streams 5: 0.22 -> 0.34 ms per 64-proof job)
    agg_streams, agg_ctxs = [], []
    if not dry and not use_dist and not args.no_secondary:
        agg_streams = [make_stream() for _ in range(AGG_JOBS_IN_FLIGHT)]
        agg_ctxs = [make_ctx(s) for s in agg_streams]
    streams = [make_stream() for _ in range(inflight)]
    ctxs = [make_ctx(s) for s in streams]
    ctx = ctxs[0]
    if inflight > 1:
        for c in ctxs:  # several MSMs in flight: the library's throughput hint (longer runs per lane; same bytes)
            c.set_throughput_hint(True)

    # Every in-flight slot works on ITS OWN input arrays (disjoint index ranges of the two seeded streams): slot 0 of
    # rank r owns [r*n, (r+1)*n) -- so an N-rank job is the MSM over [0, N*n) -- and the other slots ranges beyond N*n.
    # (Round 1 pointed all slots at the same arrays; distinct inputs keep the headline free of any cache sharing.)
    slot_first = [rank * n] + [(world + rank * (inflight - 1) + k) * n for k in range(inflight - 1)]
    d_scalars_k = [torch.empty(32 * n, dtype=torch.uint8, device=dev) for _ in range(inflight)]
    d_points_k = [torch.empty(64 * n, dtype=torch.uint8, device=dev) for _ in range(inflight)]
    dev_sync()
    for k in range(inflight):
        ctx.sample_scalars_dev(0x5EED0001, n, d_scalars_k[k].data_ptr(), first=slot_first[k])
        ctx.sample_points_dev(0x5EED0002, n, d_points_k[k].data_ptr(), first=slot_first[k])
    ctx.sync()
    d_scalars, d_points = d_scalars_k[0], d_points_k[0]
    partials = [torch.zeros(sv.G1_PARTIAL_BYTES, dtype=torch.uint8, device=dev) for _ in range(inflight)]
    gathereds = [torch.zeros(world * sv.G1_PARTIAL_BYTES, dtype=torch.uint8, device=dev) for _ in range(inflight)]
    out = torch.zeros(64, dtype=torch.uint8, device=dev)
    outs = [out] + [torch.zeros(64, dtype=torch.uint8, device=dev) for _ in range(inflight - 1)]
    dev_sync()
    step_no = [0]

    # ---- batch submission: job i of a K-step batch works on input set i (disjoint ranges again; at most 32 sets, reused
    # cyclically beyond -- 96 MiB each at 2^20); the first `inflight` sets are the slots above
    if batch:
        nsets = min(args.steps, 32)
        for k in range(inflight, nsets):
            first = (world * (1 + (inflight - 1)) + rank * (nsets - inflight) + (k - inflight)) * n
            d_scalars_k.append(torch.empty(32 * n, dtype=torch.uint8, device=dev))
            d_points_k.append(torch.empty(64 * n, dtype=torch.uint8, device=dev))
            ctx.sample_scalars_dev(0x5EED0001, n, d_scalars_k[k].data_ptr(), first=first)
            ctx.sample_points_dev(0x5EED0002, n, d_points_k[k].data_ptr(), first=first)
        ctx.sync()
        nsets = min(nsets, len(d_scalars_k))
        b_out = torch.zeros(64 * args.steps, dtype=torch.uint8, device=dev)
        job_s = [d_scalars_k[i % nsets].data_ptr() for i in range(args.steps)]
        job_p = [d_points_k[i % nsets].data_ptr() for i in range(args.steps)]

    def run_batch():
        """the K steps: one call; multi-GPU: K partials -> ONE all-gather (K x 144 B per rank) -> K folds"""
        if not use_dist:
            ctx.msm_pippenger_many_dev(job_s, job_p, [n] * args.steps, b_out.data_ptr(), args.window_bits)
        else:
            # snark-verifier_amd/distributed.py: K partials -> ONE all-gather -> K folds in one launch, on slot 0's stream
            from snark_verifier_amd.distributed import gpu_sharded_msm_batch
            res = gpu_sharded_msm_batch(ctx, [d_scalars_k[i % nsets] for i in range(args.steps)],
                                        [d_points_k[i % nsets] for i in range(args.steps)], [n] * args.steps,
                                        args.window_bits, stream=streams[0])
            run_batch.last = res

    from snark_verifier_amd.distributed import all_gather_bytes

    def step():
        k = step_no[0] % inflight
        step_no[0] += 1
        if not use_dist:
            ctxs[k].msm_pippenger_dev(d_scalars_k[k].data_ptr(), d_points_k[k].data_ptr(), n, outs[k].data_ptr(), args.window_bits)
        else:
            # snark-verifier_amd/distributed.py: shard -> HIP partial -> RCCL all-gather (144 B/rank) -> HIP fold,
            # all three enqueued on slot k's stream (every rank issues the collectives in the same order)
            with on_stream(streams[k]):
                ctxs[k].msm_pippenger_partial_dev(d_scalars_k[k].data_ptr(), d_points_k[k].data_ptr(), n,
                                                  partials[k].data_ptr(), args.window_bits)
                gathereds[k].copy_(all_gather_bytes(partials[k]))  # RCCL, or host-staged gloo: both ordered on this stream
                ctxs[k].fold_partials_dev(gathereds[k].data_ptr(), world, outs[k].data_ptr())

    def barrier():
        if use_dist:
            dist.barrier()
        dev_sync()

    # Initialisation, not a benchmark step: every in-flight slot runs the path once so that its
    # context has allocated its scratch (hipMalloc of ~1 GiB, synchronous) before anything is timed --
    # with W < in-flight slots the W warm-up steps alone would leave a slot allocating inside the timed region.
    for _ in range(inflight):
        step()
    barrier()
    step_no[0] = 0
    if batch:
        run_batch()  # initialisation again: the batch's job contexts allocate their scratch
        barrier()
        for _ in range(-(-args.warmup // args.steps)):  # W warm-up steps, rounded up to whole batches
            run_batch()
    else:
        for _ in range(args.warmup):
            step()
    barrier()

    # timed region: exactly K steps.  Per-stage HIP events are recorded on the
    # stream each kernel is launched on; they are read back after the region.
    for c in ctxs:
        c.set_stage_timing(True)
    stage_sum, stage_cnt = {}, 0
    barrier()
    t0 = time.perf_counter()
    if batch:
        run_batch()
    for i in range(0 if batch else args.steps):
        step()
        if inflight == 1:
            st = ctx.get_stage_timing()  # sequential mode: consume each step before the next
            stage_cnt += 1
            for k, v in st.items():
                stage_sum[k] = stage_sum.get(k, 0.0) + v
    barrier()
    dt = time.perf_counter() - t0
    if batch:  # [total of the call, mean k_accumulate / combine launch over the jobs' own events]
        stage_sum, stage_cnt = dict(ctx.get_stage_timing()), 1
    if stage_cnt == 0:  # pipelined mode: the events of the last step of every context that ran a timed step
        first = step_no[0] - args.steps
        used = sorted({i % inflight for i in range(first, step_no[0])})
        for c in [ctxs[k] for k in used]:
            st = c.get_stage_timing()
            stage_cnt += 1
            for k, v in st.items():
                stage_sum[k] = stage_sum.get(k, 0.0) + v
    for c in ctxs:
        c.set_stage_timing(False)
    written = list(range(inflight))  # every slot ran at least once (initialisation pass)
    out = outs[written[0]]
    slot_results = [bytes(o.cpu().numpy()) for o in outs]
    assert all(r != bytes(64) for r in slot_results) and len(set(slot_results)) == inflight  # distinct inputs, distinct sums
    if batch:  # job i of the batch = input set i: the first sets are the slots, whose single-call results are above
        got = bytes((run_batch.last if use_dist else b_out).cpu().numpy())
        jobs = [got[64 * i:64 * i + 64] for i in range(args.steps)]
        assert all(jobs[i] == slot_results[i % nsets] for i in range(args.steps) if i % nsets < inflight)
        assert len(set(jobs)) == nsets and bytes(64) not in jobs

    # single-MSM latency (strictly sequential), outside the timed region, for the record
    lat_ms, seq_stages = None, None
    if not dry:  # (local single-MSM calls: also on every rank of a multi-GPU run, outside its timed region)
        dev_sync()
        lat_out = torch.zeros(64, dtype=torch.uint8, device=dev)  # (its own buffer: `out` holds the job's result)
        # one caller, one MSM at a time: no hint
        ctx.set_throughput_hint(False)
        for _ in range(2):
            ctx.msm_pippenger_dev(d_scalars.data_ptr(), d_points.data_ptr(), n, lat_out.data_ptr(), args.window_bits)
        ctx.sync()
        t1 = time.perf_counter()
        for _ in range(8):
            ctx.msm_pippenger_dev(d_scalars.data_ptr(), d_points.data_ptr(), n, lat_out.data_ptr(), args.window_bits)
            ctx.sync()
        lat_ms = (time.perf_counter() - t1) / 8 * 1e3
        # the latency mode gives the same bytes as the hinted in-flight run (multi-GPU: slot 0's result is the fold over all ranks)
        assert use_dist or bytes(lat_out.cpu().numpy()) == slot_results[0]
        # unshared per-stage durations, for the roofline of the dominant kernel
        ctx.set_stage_timing(True)
        seq_sum = {}
        for _ in range(5):
            ctx.msm_pippenger_dev(d_scalars.data_ptr(), d_points.data_ptr(), n, lat_out.data_ptr(), args.window_bits)
            for k, v in ctx.get_stage_timing().items():  # syncs
                seq_sum[k] = seq_sum.get(k, 0.0) + v
        ctx.set_stage_timing(False)
        seq_stages = {k: v / 5 for k, v in seq_sum.items()}

    host_res = None
    if not use_dist and not dry and batch and not args.no_host_resident:
        host_res = host_resident_metrics(sv, torch, ctx, d_scalars_k, d_points_k, n, args.steps, slot_results, nsets)
    rccl_ranks_seen = None
    if use_dist:
        t = torch.tensor([dt], dtype=torch.float64)
        dist.all_reduce(t, op=dist.ReduceOp.MAX)  # (control plane)
        dt = float(t.item())
        # how many ranks the DATA plane really spans: a sum of ones over the group the partials travelled on
        ones = torch.ones(1, dtype=torch.int32, device=dev if transport["kind"] == "rccl" else "cpu")
        dist.all_reduce(ones, op=dist.ReduceOp.SUM, group=transport.get("group"))
        rccl_ranks_seen = int(ones.item())
    result_hex = bytes(out.cpu().numpy()).hex()

    # BASELINE config 4 (2^24 points IN TOTAL, strong scaling) in the same run, on every world size
    strong_res = None
    if not strong and not args.no_strong and batch and (not dry or args.strong_total_log2n < 20):
        for c in ctxs[1:]:
            c.sync()
        strong_res = strong_leg(torch, dist, sv, ctx, streams[0], world, rank, args.strong_total_log2n, max(2, min(args.steps, 5)),
                                args.window_bits, dry, dev, dev_sync)

    line = None
    if rank == 0:
        stages = {k: v / stage_cnt for k, v in stage_sum.items()}
        dom = max((k for k in stages if k != "total"), key=lambda k: stages[k])
        dom_ms = stages[dom]
        launch_n = launch_points(n, args.window_bits)  # n, or the 2^20-point chunk large MSMs are pipelined in
        # the dominant kernel's per-launch duration: ONE launch alone on the GPU (sequential single-MSM calls, HIP events on
        # the context's stream).  In the timed batch ~1.5 launches of it are co-resident, so an in-batch launch lasts longer
        # than ms_per_step -- that stretched figure is kept as `frac_in_batch`; it is not the kernel's speed.
        dom_alone_ms = (seq_stages or {}).get(dom, 0.0) or dom_ms
        achieved = BYTES_PER_POINT * launch_n / (dom_alone_ms * 1e-3) / 1e9
        achieved_in_batch = BYTES_PER_POINT * launch_n / (dom_ms * 1e-3) / 1e9
        line = {
            "metric": "BN254 G1 MSM points/sec at 2^%d" % (args.total_log2n if strong else args.log2n),
            "value": world * n * args.steps / dt,
            "unit": "points/s",
            "n_gpus": world,
            "steps": args.steps,
            "warmup": args.warmup,
            "ms_per_step": dt / args.steps * 1e3,
            "higher_is_better": True,
            "scaling": "strong" if strong else "weak",
            "vs_baseline": None,
            "dtype": "i32x9 (254-bit Montgomery Fq as 9 x 29-bit signed lazy limbs on the integer VALU, 64-bit column accumulators)",
            "data": "synthetic" if not dry else "synthetic; DRY RUN on CPU doubles of the device context (a control-flow test, not a measurement)",
            "config": {
                "workload": ("BN254 G1 Pippenger MSM, 2^%d random points/scalars IN TOTAL sharded over %d GPU(s) (2^%d each), "
                             "inputs resident in HBM, affine result (configs[3])" % (args.total_log2n, world, args.log2n)) if strong
                            else "BN254 G1 Pippenger MSM, 2^%d random points/scalars per GPU, inputs resident in HBM, "
                                 "affine result (configs[1])" % args.log2n,
                "points_per_gpu": n,
                "points_per_kernel_launch": launch_n,
                "window_bits": args.window_bits or "default",
                "parallelism": ("point-sharded x%d: per batch ONE all-gather of %d x 144 B partials per rank, then the %d folds in one "
                                "launch" % (world, args.steps, args.steps)) if batch else
                               "point-sharded x%d, all-gather of 144 B partials + local fold" % world,
                "submission": ("the %d timed steps are ONE snarkv_g1_msm_pippenger_many_dev call (the library pipelines the "
                               "MSMs: sorts on high-priority streams, accumulations back to back, one batched tail)" % args.steps)
                              if batch else "%d single-MSM calls kept in flight, one context + stream each" % inflight,
                "msms_in_flight": args.steps if batch else inflight,
                "inputs": ("job i works on input set i mod %d, every set its own scalar / point arrays (disjoint ranges of the "
                           "seeded streams)" % nsets) if batch else
                          "every in-flight slot has its own scalar / point arrays (disjoint ranges of the seeded streams)",
                "throughput_hint": batch or inflight > 1,  # runs of 96 entries per lane instead of 64 (snarkv_ctx_set_throughput_hint on
                                                           # in-flight contexts; always on a batch's jobs); the single-MSM latency and
                                                           # the sequential stage times are taken without it
                "single_msm_latency_ms": lat_ms,  # ONE util::msm::multi_scalar_multiplication call at a time (msm.rs:308), host wall
                "single_msm_points_per_s": (n / (lat_ms * 1e-3)) if lat_ms else None,
                "rccl_ranks_seen": rccl_ranks_seen if transport["kind"] == "rccl" else None,  # sum of ones over the RCCL group
                "ranks_share_devices": (not dry) and shared_devices,  # true only on a test box with fewer GPUs than ranks
                "data_plane_ranks_seen": rccl_ranks_seen,  # ... over whatever group the partials travelled on (null: one process)
                "transport": None if not use_dist else {
                    "kind": "RCCL all-gather over xGMI (torch.distributed backend nccl)" if transport["kind"] == "rccl"
                            else "gloo all-gather, HOST-STAGED (partials copied to the host and back)",
                    "requested": transport.get("requested"), "fallback_reason": transport.get("why"),
                    "probe_seconds": transport.get("probe_seconds"), "control_plane": "gloo",
                    "rccl_env": {k: os.environ.get(k) for k in ("NCCL_SOCKET_IFNAME", "NCCL_IB_DISABLE", "HSA_ENABLE_IPC_MODE_LEGACY")}},
                "result": result_hex,
            },
            "roofline": {
                "bound": "hbm",
                "kernel": "stage '%s'" % dom,
                "achieved": achieved,
                "peak": HBM_PEAK_GBPS,
                "unit": "GB/s",
                "frac": achieved / HBM_PEAK_GBPS,
                "kernel_ms": dom_alone_ms,
                "achieved_in_batch": achieved_in_batch,
                "frac_in_batch": achieved_in_batch / HBM_PEAK_GBPS,
                "kernel_ms_in_batch": dom_ms,
                # HBM-side bytes per launch of the dominant kernel from the rocprofv3 PMC record of THIS tree's kernels
                # (tools/pmc_traffic.py -> profiles/r*_pmc_hbm_traffic*.json, keyed by the kernel-source hash); null when
                # no record matches the sources.
                "traffic": None,  # filled below from the PMC record of THESE kernels, or left null

                "note": "algorithmic 96 B/point x %d points per launch / the dominant kernel's avg HIP-event duration with ONE "
                        "launch on the GPU (`kernel_ms`, <= ms_per_step); `*_in_batch`: the same launch inside the timed region, "
                        "stretched by the %s launches of its kind that share the GPU there.  The path is integer-VALU-bound "
                        "(~160 Fq products per point), see DESIGN.md section 4" % (launch_n, "~1.5" if batch else str(inflight)),
            },
            "stages_ms": stages,
        }
        # HBM-side traffic of the dominant kernel: rocprofv3 PMC (tools/pmc_traffic.py: separate FETCH_SIZE / WRITE_SIZE
        # passes, gfx950 correction 2 x FETCH + WRITE), quoted only from a record taken on the kernels of this tree
        pmc, where = _profile_record("r*_pmc_hbm_traffic%s.json" % ("" if args.log2n == 20 else "_2p%d" % args.log2n))
        kname = {"bucket_accumulate": "k_accumulate"}.get(dom)
        if pmc and kname in pmc["kernels"]:
            k = pmc["kernels"][kname]
            line["roofline"]["traffic"] = k["bytes_corrected"]
            line["roofline"]["traffic_as_counted"] = k["bytes_as_counted"]
            line["roofline"]["traffic_fetch_factor"] = k.get("fetch_factor")
            line["roofline"]["traffic_source"] = where + " (kernel_source_hash %s)" % pmc["kernel_source_hash"]
            # the kernel's own byte accounting: one 64-byte point gather + one 8-byte entry per (half-scalar, window), the
            # run partials and the bucket grid written once
            Wn = -(-128 // 16) if not args.window_bits else -(-128 // args.window_bits)
            entries = 2 * launch_n * Wn
            model = entries * 72 + (entries // 96 + 1) * 288 + Wn * (1 << 15) * 144
            line["roofline"]["traffic_model_bytes"] = model
            line["roofline"]["traffic_over_algorithmic"] = k["bytes_corrected"] / (BYTES_PER_POINT * launch_n)
            line["roofline"]["traffic_note"] = (
                "per launch of k_accumulate, fabric side of L2 (Infinity-Cache hits included).  FETCH_SIZE on gfx950 tallies the "
                "128-byte requests of wide coalesced reads at 64 B (x2, calibrated on k_prepare's 96 B/point stream: see the "
                "record) but counts this kernel's 64-byte point gathers as they are (x1): the figure agrees with the byte "
                "model (%d entries x (64 + 8) B + partials + grid = %.2f GB) within ~20 %%.  ~%.0fx the 96 B/point input: every "
                "Montgomery point is gathered once per window -- inherent to bucket accumulation, not re-reads of the input"
                % (entries, model / 1e9, k["bytes_corrected"] / (BYTES_PER_POINT * launch_n)))
        else:
            line["roofline"]["traffic_source"] = where if not pmc else "record has no %s" % kname
        # the same kernel's average duration in the tracked rocprofv3 summary (tools/collect_profiles.sh writes the CSV and a
        # sidecar with the kernel-source hash): `kernel_ms` above is THIS run's HIP events, `kernel_us_profile` the profile's
        prof, pwhere = _profile_record("r*_kernel_stats_sequential.json")
        if prof and kname in prof.get("kernels", {}):
            pk = prof["kernels"][kname]
            line["roofline"]["kernel_us_profile"] = pk["avg_us"]
            line["roofline"]["kernel_us_profile_calls"] = pk.get("calls")
            line["roofline"]["kernel_us_profile_source"] = "%s <- %s (kernel_source_hash %s)" % (pwhere, prof.get("csv"), prof["kernel_source_hash"])
            line["roofline"]["frac_from_profile"] = BYTES_PER_POINT * launch_n / (pk["avg_us"] * 1e-6) / 1e9 / HBM_PEAK_GBPS
        else:
            line["roofline"]["kernel_us_profile"] = None
            line["roofline"]["kernel_us_profile_source"] = pwhere if not prof else "record has no %s" % kname
        # issue roofline of the dominant kernel from its ISA (tools/isa_stats.py), same hash rule
        isa, iwhere = _profile_record("r*_isa_k_accumulate.json")
        if isa:
            line["issue_roofline"] = {
                "kernel": "k_accumulate", "mads_per_entry": isa["mads_per_entry"], "instr_per_entry": isa["instr_per_entry"],
                "frac": isa["useful_issue_fraction"], "source": iwhere,
                "note": "static ISA census of the accumulate loop: v_mad_i64_i32 (the 29 x 29-bit partial products) per issued "
                        "instruction; on gfx950 every instruction of this mix costs one ~4.2-cycle issue slot "
                        "(profiles/r02_ubench_issue.txt), so this is the fraction of issue slots doing arithmetic that the "
                        "algorithm needs",
            }
        else:
            line["issue_roofline"] = {"kernel": "k_accumulate", "frac": None, "source": iwhere}
        if not use_dist:
            # Integer-VALU roofline (the binding constraint): the same
            # mixed addition as k_accumulate in isolation, 4 waves/SIMD on every CU.
            W = -(-128 // 16) if not args.window_bits else -(-128 // args.window_bits)
            adds = 2.0 * launch_n * W  # one bucket addition per (half-scalar, window) of the points one launch covers
            peak_madd = ctx.ubench_valu(1, 300)
            peak_mul = ctx.ubench_valu(0, 2000)
            acc_ms = (seq_stages or stages).get("bucket_accumulate", 0.0)
            if acc_ms > 0:
                line["valu_roofline"] = {
                    "kernel": "k_accumulate", "unit": "G1 mixed additions/s",
                    "achieved": adds / (acc_ms * 1e-3), "peak": peak_madd,
                    "frac": adds / (acc_ms * 1e-3) / peak_madd,
                    "peak_fq_mul_per_s": peak_mul,
                    "note": "peak = xyzz29_madd_fast chains with no memory traffic, measured live "
                            "(snarkv_ubench_valu); achieved uses the unshared (sequential) launch duration",
                }
        if inflight > 1 and dom_ms > 0:
            # launches of the dominant kernel overlap in the timed region (that overlap is what the in-flight mode is for), so a
            # launch's duration stretches with the number of its kind resident at once: sum of launch durations / wall time
            conc = dom_ms * args.steps / (dt * 1e3)
            line["roofline"]["launches_resident_on_average"] = conc
            line["roofline"]["overlap_note"] = (
                "with %s %.2f launches of this kernel share the GPU on average in the timed region, so an in-batch launch lasts "
                "%.2f ms (> ms_per_step) -- `frac_in_batch`; `frac` uses the launch alone (%.3f ms)"
                % ("a batch's accumulations on three streams" if batch else "%d MSMs in flight" % inflight, conc, dom_ms, dom_alone_ms))
        if seq_stages:
            dseq = seq_stages.get(dom, 0.0)
            line["stages_ms_sequential"] = seq_stages
            if dseq > 0:  # (kept under the round-2 / round-3 names too: the same numbers as `achieved` / `frac`)
                line["roofline"]["achieved_unshared"] = BYTES_PER_POINT * launch_n / (dseq * 1e-3) / 1e9
                line["roofline"]["frac_unshared"] = line["roofline"]["achieved_unshared"] / HBM_PEAK_GBPS
        if not args.no_cpu_baseline and world == 1:
            cb, cpu_out, (s, p) = cpu_baseline(ctx, d_scalars, d_points, min(args.cpu_sample_log2 or args.log2n, args.log2n))
            # the GPU must agree with the CPU restatement on that same sample
            m = len(s) // 32
            chk = torch.zeros(64, dtype=torch.uint8, device="cuda")
            torch.cuda.synchronize()
            ctx.msm_pippenger_dev(d_scalars.data_ptr(), d_points.data_ptr(), m, chk.data_ptr(), 0)
            ctx.sync()
            cb["gpu_matches_on_sample"] = bytes(chk.cpu().numpy()) == cpu_out
            cb["sample_is_the_whole_workload"] = m == n
            line["cpu_baseline"] = cb
        if host_res:
            line["host_resident"] = host_res
        if not use_dist and not args.no_secondary:
            line["secondary"] = secondary_metrics(sv, torch, agg_ctxs, cpu=not args.no_cpu_baseline)
            line["secondary"]["aggregate_64_proofs_context_free_16_threads"] = context_free_metrics(sv)
            line["secondary"]["aggregate_1024_proofs_context_free_16_threads"] = context_free_metrics(sv, nproofs=1024)
            e2e = end_to_end_metrics()
            if e2e:
                line["secondary"].update(e2e)
            # LAST key of the line (a log tail keeps it): the named BASELINE configs, ONE job at a time, then the others
            sec = line["secondary"]
            line["named_configs"] = {
                "msm_2p%d_points_per_s" % args.log2n: line["value"],
                "aggregate_64_proofs_one_job_ms": sec.get("aggregate_64_proofs", {}).get("ms"),
                "aggregate_1024_proofs_one_job_ms": sec.get("aggregate_1024_proofs", {}).get("ms"),
                "decide_all_1_ms": sec.get("decide_all_1", {}).get("ms"),
                # config 5 (1 024 distinct proofs from proof bytes to the verdict, median of 25 calls): the reference example's own
                # transcripts first (Poseidon for the snarks and the accumulation proof), the Keccak route beside it
                "end_to_end_aggregate_1024_distinct_proofs_poseidon_ms": sec.get("end_to_end_aggregate_1024_distinct_proofs_poseidon", {}).get("ms"),
                "end_to_end_aggregate_1024_distinct_proofs_keccak_ms": sec.get("end_to_end_aggregate_1024_distinct_proofs", {}).get("ms"),
                "end_to_end_aggregate_64_proofs_bdfg21_poseidon_ms": sec.get("end_to_end_aggregate_64_proofs_bdfg21_poseidon", {}).get("ms"),
                "host_resident_points_per_s": (host_res or {}).get("value"),
                "NOT_one_job (16 jobs in flight / merged, proofs per s)": {
                    "aggregate_64_proofs_pipelined": sec.get("aggregate_64_proofs_pipelined", {}).get("proofs_per_s"),
                    "aggregate_64_proofs_context_free_16_threads": sec.get("aggregate_64_proofs_context_free_16_threads", {}).get("proofs_per_s"),
                    "aggregate_64_proofs_merged": sec.get("aggregate_64_proofs_merged", {}).get("proofs_per_s"),
                    "aggregate_1024_proofs_pipelined": sec.get("aggregate_1024_proofs_pipelined", {}).get("proofs_per_s"),
                    "aggregate_1024_proofs_context_free_16_threads": sec.get("aggregate_1024_proofs_context_free_16_threads", {}).get("proofs_per_s"),
                    "aggregate_1024_proofs_merged": sec.get("aggregate_1024_proofs_merged", {}).get("proofs_per_s")},
            }
        if use_dist and not dry and world > 1 and batch:
            # the N-GPU result against ONE GPU: rank 0 samples job 0's points of ALL ranks ([0, N n) of the seeded streams)
            # and reduces them alone
            s_all = torch.empty(32 * n * world, dtype=torch.uint8, device=dev)
            p_all = torch.empty(64 * n * world, dtype=torch.uint8, device=dev)
            o_all = torch.zeros(64, dtype=torch.uint8, device=dev)
            dev_sync()
            ctx.sample_scalars_dev(0x5EED0001, n * world, s_all.data_ptr(), first=0)
            ctx.sample_points_dev(0x5EED0002, n * world, p_all.data_ptr(), first=0)
            ctx.msm_pippenger_dev(s_all.data_ptr(), p_all.data_ptr(), n * world, o_all.data_ptr(), args.window_bits)
            ctx.sync()
            line["config"]["result_matches_one_gpu_recompute"] = bytes(o_all.cpu().numpy()).hex() == result_hex
            del s_all, p_all
        if strong_res is not None:
            line["config4_strong"] = strong_res

    if use_dist:
        dist.barrier()
        dist.destroy_process_group()
    if rank == 0:
        if not dry and not args.no_mgpu_leg:
            # the single-process form of the same batch (snarkv_mgpu_*), in a process of its own, after the ranks have left
            if use_dist:
                time.sleep(2.0)  # the other ranks' processes release their GPUs
            line["single_process_mgpu"] = run_mgpu_leg_subprocess(world, args.steps, args.log2n, args.window_bits)
        if "named_configs" in line:  # keep it the LAST key of the line
            line["named_configs"] = line.pop("named_configs")
        emit(line)
    if transport.get("probe_hung"):
        sys.stdout.flush()
        os._exit(0)  # an abandoned RCCL probe thread must not hold the interpreter at shutdown


if __name__ == "__main__":
    main()
