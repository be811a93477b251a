"""GPU: two REAL ranks before the driver finds a multi-GPU node.  Two OS processes, each with its own HIP
context and streams on the box's one device, run the product's multi-process wiring (snark-verifier_amd/distributed.py over
the C ABI) on real kernels: K point-sharded MSMs with ONE exchange of K x 144 B per rank and the K folds on every rank, the
single-MSM and bucket-sharded forms, and the proof-sharded aggregation incl. a shard that rejects.  The exchange is
host-staged over gloo (RCCL refuses one device twice); everything else is what an N-GPU run executes per rank."""
import json
import os
import subprocess
import sys

import pytest

import coracle as C

pytestmark = pytest.mark.gpu
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def test_two_processes_one_device():
    world = 2
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", str(world), "--master-addr",
           "127.0.0.1", "--master-port", str(37000 + os.getpid() % 1000), os.path.join(ROOT, "tests", "two_rank_worker.py")]
    env = dict(os.environ, HSA_ENABLE_IPC_MODE_LEGACY="0")
    r = subprocess.run(cmd, capture_output=True, text=True, timeout=900, cwd=ROOT, env=env)
    assert r.returncode == 0, (r.stdout[-2000:], r.stderr[-4000:])
    # (the two ranks share one stdout pipe: decode at every marker, wherever the other rank's bytes landed around it)
    dec, lines, pos = json.JSONDecoder(), [], 0
    while (pos := r.stdout.find("RANKLINE ", pos)) >= 0:
        pos += len("RANKLINE ")
        lines.append(dec.raw_decode(r.stdout, pos)[0])
    assert sorted(d["rank"] for d in lines) == list(range(world)) and len({d["pid"] for d in lines}) == world
    totals = [200_000, 4097, 1]
    exp = b"".join(C.msm_pippenger(C.sample_scalars(0x7A00 + i, n), C.sample_points(0x7B00 + i, n), 8) for i, n in enumerate(totals))
    for d in lines:  # every rank holds the same K results: the oracle's
        assert bytes.fromhex(d["batch"]) == exp, d["rank"]
        assert bytes.fromhex(d["single"]) == exp[:64] and bytes.fromhex(d["bucket_sharded"]) == exp[:64], d["rank"]
        assert d["agg_ok"] and d["agg_acc_matches_fixture"] and d["agg_bad"], d
    # ... and the single-process MSM over all the points gives the same bytes
    import snark_verifier_amd as sv

    ctx = sv.Context(0)
    assert ctx.msm_pippenger(C.sample_scalars(0x7A00, totals[0]), C.sample_points(0x7B00, totals[0])) == exp[:64]
    ctx.close()
