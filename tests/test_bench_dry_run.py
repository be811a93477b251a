"""CPU: bench.py's N > 1 control flow before the driver ever has a multi-GPU node -- launched exactly
as the driver launches it (`python -m torch.distributed.run --nproc-per-node N ... bench.py --gpus N --steps K --warmup W`),
with `--dry-run-doubles tests/bench_doubles.py` putting oracle-backed CPU doubles in place of the device context and gloo in
place of RCCL.  What is checked: the rendezvous, the disjoint shards, the batch submission + ONE all-gather + folds, the
max-over-ranks timing, and the single JSON line rank 0 prints: metric / unit / n_gpus / steps / scaling / the aggregate value
N * n * K / t / config labels -- and that the result it reports equals the single-process MSM over all ranks' points."""
import os
import subprocess
import sys
import tempfile

import pytest

import bench_line
import coracle as C

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _run(world, extra, port, launcher=True):
    cmd = [sys.executable]
    if launcher:
        cmd += ["-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", str(world), "--master-addr", "127.0.0.1",
                "--master-port", str(port)]
    cmd += [os.path.join(ROOT, "bench.py"), "--gpus", str(world), "--dry-run-doubles", os.path.join(ROOT, "tests", "bench_doubles.py")] + extra
    env = {k: v for k, v in os.environ.items() if k not in ("WORLD_SIZE", "RANK", "LOCAL_RANK")}
    with tempfile.TemporaryDirectory() as tmp:
        env["SNARKV_BENCH_DETAILS"] = os.path.join(tmp, "details.json")
        r = subprocess.run(cmd, capture_output=True, text=True, timeout=600, cwd=ROOT, env=env)
        assert r.returncode == 0, r.stderr[-3000:]
        # ONE compact line (< 4 KB, the last of stdout: what the driver parses); the tests below read the full record it names
        compact, full = bench_line.parse(r.stdout)
    assert compact["config"]["points_per_gpu"] == full["config"]["points_per_gpu"]
    full["_compact"] = compact
    return full


@pytest.mark.parametrize("world", [2, 3])
def test_weak_scaling_line_from_n_ranks(world):
    log2n, steps = 8, 3
    d = _run(world, ["--steps", str(steps), "--warmup", "1", "--log2n", str(log2n)], 36000 + world + os.getpid() % 1000)
    n = 1 << log2n
    assert d["metric"] == "BN254 G1 MSM points/sec at 2^%d" % log2n and d["unit"] == "points/s"
    assert d["n_gpus"] == world and d["steps"] == steps and d["warmup"] == 1
    assert d["scaling"] == "weak" and d["higher_is_better"] is True and d["vs_baseline"] is None
    assert d["config"]["points_per_gpu"] == n and d["config"]["msms_in_flight"] == steps
    assert abs(d["value"] - world * n * steps / (d["ms_per_step"] * steps * 1e-3)) < 1e-6 * d["value"]  # whole-job aggregate
    assert "DRY RUN" in d["data"] and "cpu_baseline" not in d and "secondary" not in d
    assert "point-sharded x%d" % world in d["config"]["parallelism"]
    # the sum of ones over the group the partials travelled on: it spans every rank -- gloo here, so the RCCL count stays null
    assert d["config"]["data_plane_ranks_seen"] == world and d["config"]["rccl_ranks_seen"] is None
    assert "HOST-STAGED" in d["config"]["transport"]["kind"] and d["config"]["transport"]["control_plane"] == "gloo"
    # slot 0 of rank r holds points [r n, (r + 1) n) of the seeded streams: the reported result is the MSM over [0, N n)
    s, p = C.sample_scalars(0x5EED0001, world * n), C.sample_points(0x5EED0002, world * n)
    assert d["config"]["result"] == C.msm_pippenger(s, p, 4).hex()


def test_strong_scaling_line_config4_shape():
    d = _run(2, ["--steps", "2", "--warmup", "1", "--total-log2n", "10"], 37000 + os.getpid() % 1000)
    assert d["scaling"] == "strong" and d["metric"].endswith("2^10") and d["config"]["points_per_gpu"] == 512
    assert "configs[3]" in d["config"]["workload"]
    s, p = C.sample_scalars(0x5EED0001, 1024), C.sample_points(0x5EED0002, 1024)
    assert d["config"]["result"] == C.msm_pippenger(s, p, 4).hex()


def test_inflight_mode_under_dist():
    """`--inflight 2`: single-MSM calls (partial -> all-gather -> fold per step) instead of the batch"""
    d = _run(2, ["--steps", "3", "--warmup", "1", "--log2n", "7", "--inflight", "2"], 38000 + os.getpid() % 1000)
    assert d["n_gpus"] == 2 and d["config"]["msms_in_flight"] == 2
    s, p = C.sample_scalars(0x5EED0001, 256), C.sample_points(0x5EED0002, 256)
    assert d["config"]["result"] == C.msm_pippenger(s, p, 4).hex()


def test_plain_launch_without_torchrun_starts_the_ranks_itself():
    """`python bench.py --gpus 2 ...` with WORLD_SIZE unset must MEASURE, not exit: it launches the two
    ranks under torch.distributed.run itself and relays rank 0's one line, with the launch attempts recorded."""
    log2n, steps = 8, 2
    d = _run(2, ["--steps", str(steps), "--warmup", "1", "--log2n", str(log2n)], 0, launcher=False)
    assert d["n_gpus"] == 2 and d["config"]["data_plane_ranks_seen"] == 2
    la = d["config"]["launch"]
    assert la["self_launched"] is True and la["attempts"][0]["rc"] == 0 and len(la["attempts"]) == 1
    assert d["_compact"]["config"]["launch"] == {"self_launched": True, "attempts": 1, "how": la["attempts"][0]["how"][:80]}
    s, p = C.sample_scalars(0x5EED0001, 2 << log2n), C.sample_points(0x5EED0002, 2 << log2n)
    assert d["config"]["result"] == C.msm_pippenger(s, p, 4).hex()


def test_config4_strong_leg_rides_in_the_same_run():
    """the weak line and BASELINE config 4 (2^k points IN TOTAL over the ranks) come out of ONE run"""
    d = _run(2, ["--steps", "2", "--warmup", "1", "--log2n", "7", "--strong-total-log2n", "9"], 39000 + os.getpid() % 1000)
    assert d["scaling"] == "weak"
    c4 = d["config4_strong"]
    assert c4["scaling"] == "strong" and c4["n_gpus"] == 2 and c4["points_per_gpu"] == 256
    assert abs(c4["value"] - 512 * c4["steps"] / (c4["ms_per_step"] * c4["steps"] * 1e-3)) < 1e-6 * c4["value"]
    s, p = C.sample_scalars(0x5EED0011, 512), C.sample_points(0x5EED0012, 512)
    assert c4["result"] == C.msm_pippenger(s, p, 4).hex()


def test_rccl_probe_failure_makes_every_rank_fall_back_to_gloo():
    """`--transport auto` on a box where RCCL cannot come up (here: no GPU at all) -- every rank's probe
    fails, the ranks agree over the gloo control plane, the partials travel host-staged, and the line carries the reason."""
    d = _run(2, ["--steps", "2", "--warmup", "1", "--log2n", "7", "--transport", "auto", "--no-strong"], 40000 + os.getpid() % 1000)
    tr = d["config"]["transport"]
    assert "HOST-STAGED" in tr["kind"] and tr["requested"] == "auto" and "RCCL unavailable on rank(s) [0, 1]" in tr["fallback_reason"]
    assert d["config"]["data_plane_ranks_seen"] == 2 and d["config"]["rccl_ranks_seen"] is None
    s, p = C.sample_scalars(0x5EED0001, 256), C.sample_points(0x5EED0002, 256)
    assert d["config"]["result"] == C.msm_pippenger(s, p, 4).hex()


def test_the_contract_line_never_outgrows_the_drivers_log_tail(tmp_path, capsys, monkeypatch):
    """Round 5's line was 26.5 KB, the driver's log tail kept ~8 KB of it and the round went unmeasured.  `bench.emit`
    prints an extract of the record under 4 KB and, should a future key push it over, drops the OPTIONAL parts (and says
    so in the line) instead of failing or growing -- the contract keys, `roofline` and `cpu_baseline` stay."""
    import importlib.util
    import json

    spec = importlib.util.spec_from_file_location("_bench_mod", os.path.join(ROOT, "bench.py"))
    bench = importlib.util.module_from_spec(spec)
    spec.loader.exec_module(bench)
    monkeypatch.setenv("SNARKV_BENCH_DETAILS", str(tmp_path / "details.json"))
    line = {"metric": "m", "value": 1.0, "unit": "points/s", "n_gpus": 1, "steps": 1, "warmup": 0, "ms_per_step": 1.0, "higher_is_better": True,
            "scaling": "weak", "vs_baseline": None, "data": "synthetic", "dtype": "x" * 5000,
            "config": {"workload": "w", "points_per_gpu": 1, "notes": "n" * 20000},
            "roofline": {"bound": "hbm", "kernel": "k", "achieved": 1.0, "peak": 8000.0, "unit": "GB/s", "frac": 0.1, "traffic": None, "note": "z" * 9000},
            "cpu_baseline": {"value": 1.0, "unit": "points/s", "cores": 1, "kind": "port", "sample": "s" * 3000},
            "stages_ms": {("stage%d" % i): 0.1 * i for i in range(300)},
            "named_configs": {("config_%d" % i): 1.23456789 * i for i in range(400)}}
    bench.emit(line)
    out = [ln for ln in capsys.readouterr().out.splitlines() if ln.strip()]
    assert len(out) == 1 and len(out[0]) < bench.LINE_LIMIT == 4096
    c = json.loads(out[0])
    assert c["dropped_for_size"] == ["named_configs"] and c["named_configs"] == "see details"
    for k in ("metric", "value", "unit", "n_gpus", "steps", "warmup", "ms_per_step", "higher_is_better", "scaling", "vs_baseline", "dtype", "data"):
        assert k in c
    assert c["roofline"]["frac"] == 0.1 and "note" not in c["roofline"] and c["cpu_baseline"]["kind"] == "port"
    assert len(c["cpu_baseline"]["sample"]) <= 260 and "notes" not in c["config"] and "stages_ms" not in c
    with open(str(tmp_path / "details.json")) as f:
        assert json.load(f) == line  # the whole record is in the file the line names
    assert c["details"] == str(tmp_path / "details.json")
