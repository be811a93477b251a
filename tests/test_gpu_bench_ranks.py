"""GPU: `bench.py --gpus 2` for real, before the driver finds a multi-GPU node: a PLAIN launch (no
torchrun) on the box's one device.  bench.py starts the two ranks itself; both land on device 0 (`ranks_share_devices`), so
every rank's RCCL probe fails -- two ranks of one communicator on one device is exactly what RCCL refuses -- the ranks agree
over the gloo control plane, and the partials travel host-staged: the whole fallback path on hardware, with real kernels,
checked against the C oracle.  In the same line: BASELINE config 4 in shape (2^k points IN TOTAL over the ranks), the
one-GPU recompute of the sharded result, and the single-process `snarkv_mgpu_*` leg with two ranks."""
import os
import subprocess
import sys

import pytest

import bench_line
import coracle as C

pytestmark = pytest.mark.gpu
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


@pytest.mark.parametrize("ranks", [2, 4])
def test_plain_launch_n_ranks_one_device_measures_and_says_how(tmp_path, ranks):
    """First-contact hardening for N > 1 without the hardware: N OS processes on the box's one device
    through a PLAIN `python bench.py --gpus N`; the compact line is the last line, says how the partials travelled, the
    N-GPU results equal the one-GPU recomputes, and the whole thing is done within two minutes."""
    import time

    log2n, steps = 14, 3
    t_start = time.perf_counter()
    env = {k: v for k, v in os.environ.items() if k not in ("WORLD_SIZE", "RANK", "LOCAL_RANK")}
    env["SNARKV_BENCH_RCCL_PROBE_TIMEOUT"] = "120"
    cmd = [sys.executable, os.path.join(ROOT, "bench.py"), "--gpus", str(ranks), "--steps", str(steps), "--warmup", "1", "--log2n", str(log2n),
           "--strong-total-log2n", "16", "--no-cpu-baseline", "--no-secondary"]
    env["SNARKV_BENCH_DETAILS"] = os.path.join(str(tmp_path), "details.json")
    r = subprocess.run(cmd, capture_output=True, text=True, timeout=1500, cwd=ROOT, env=env)
    assert r.returncode == 0, r.stderr[-4000:]
    compact, d = bench_line.parse(r.stdout)
    # what the driver reads off the line itself on its first N > 1 contact
    assert time.perf_counter() - t_start < 120.0, "the N-rank launch took %.0f s" % (time.perf_counter() - t_start)
    assert compact["n_gpus"] == ranks and compact["config"]["launch"]["self_launched"] is True
    assert compact["config"]["ranks_share_devices"] is True and compact["config"]["data_plane_ranks_seen"] == ranks
    assert "HOST-STAGED" in compact["config"]["transport"]["kind"] and "RCCL unavailable" in compact["config"]["transport"]["fallback_reason"]
    assert compact["config4_strong"]["matches_one_gpu_single_call"] is True and compact["config"]["result_matches_one_gpu_recompute"] is True
    assert compact["single_process_mgpu"]["job0_matches_one_gpu_recompute"] is True and compact["single_process_mgpu"]["peer_copy"] > 0
    n = 1 << log2n
    assert d["n_gpus"] == ranks and d["scaling"] == "weak" and d["config"]["points_per_gpu"] == n
    assert abs(d["value"] - ranks * n * steps / (d["ms_per_step"] * steps * 1e-3)) < 1e-6 * d["value"]
    cfg = d["config"]
    assert cfg["launch"]["self_launched"] is True and cfg["launch"]["attempts"][0]["rc"] == 0
    assert cfg["ranks_share_devices"] is True
    tr = cfg["transport"]
    assert "HOST-STAGED" in tr["kind"] and tr["requested"] == "auto" and "RCCL unavailable on rank(s)" in tr["fallback_reason"]
    assert cfg["rccl_ranks_seen"] is None and cfg["data_plane_ranks_seen"] == ranks
    # the sharded result: the MSM over ALL ranks' points, by the C oracle and by rank 0's own one-GPU recompute
    s, p = C.sample_scalars(0x5EED0001, ranks * n), C.sample_points(0x5EED0002, ranks * n)
    assert cfg["result"] == C.msm_pippenger(s, p, 8).hex() and cfg["result_matches_one_gpu_recompute"] is True
    c4 = d["config4_strong"]
    assert c4["n_gpus"] == ranks and c4["points_per_gpu"] == (1 << 16) // ranks and c4["matches_one_gpu_single_call"] is True
    s, p = C.sample_scalars(0x5EED0011, 1 << 16), C.sample_points(0x5EED0012, 1 << 16)
    assert c4["result"] == C.msm_pippenger(s, p, 8).hex()
    # the single-process leg: N ranks on device 0 -- RCCL refuses the duplicate LOUDLY, peer copies deliver
    mg = d["single_process_mgpu"]
    assert mg["ranks"] == ranks and mg["devices"] == [0] * ranks
    assert "distinct devices" in mg["rccl"]["error"] and mg["peer_copy"]["value"] > 0
    assert mg["job0_matches_one_gpu_recompute"] is True and mg["result_job0"] == cfg["result"]


def test_rccl_data_plane_at_world_one_and_the_line_is_the_last_line(tmp_path):
    """`--force-dist` on one rank: gloo control plane + a REAL RCCL group for the partials (a communicator of one), and the
    JSON line is the LAST line of stdout -- RCCL prints a version banner through C stdio that used to land after it."""
    env = {k: v for k, v in os.environ.items() if k not in ("WORLD_SIZE", "RANK", "LOCAL_RANK")}
    env["MASTER_PORT"] = str(41000 + os.getpid() % 1000)
    cmd = [sys.executable, os.path.join(ROOT, "bench.py"), "--force-dist", "--steps", "3", "--warmup", "1", "--log2n", "14", "--no-cpu-baseline",
           "--no-secondary", "--no-strong", "--no-mgpu-leg"]
    env["SNARKV_BENCH_DETAILS"] = os.path.join(str(tmp_path), "details.json")
    r = subprocess.run(cmd, capture_output=True, text=True, timeout=900, cwd=ROOT, env=env)
    assert r.returncode == 0, r.stderr[-3000:]
    compact, d = bench_line.parse(r.stdout)  # (asserts that the line is the last one)
    assert compact["config"]["rccl_ranks_seen"] == 1 and "RCCL all-gather" in compact["config"]["transport"]["kind"]
    tr = d["config"]["transport"]
    assert "RCCL all-gather" in tr["kind"] and tr["fallback_reason"] is None and d["config"]["rccl_ranks_seen"] == 1
    assert tr["rccl_env"]["NCCL_SOCKET_IFNAME"] == os.environ.get("NCCL_SOCKET_IFNAME", "lo")
    n = 1 << 14
    assert d["config"]["result"] == C.msm_pippenger(C.sample_scalars(0x5EED0001, n), C.sample_points(0x5EED0002, n), 8).hex()


def test_the_default_line_keeps_the_drivers_contract(tmp_path):
    """`python bench.py` (N = 1, small size here): ONE JSON line, the last of stdout, with every key the driver's contract
    names -- metric / value / unit / n_gpus / steps / warmup / ms_per_step / higher_is_better / scaling / vs_baseline / dtype /
    data / config.workload (no model keys), `roofline` {bound, achieved, peak, unit, frac, traffic} and `cpu_baseline`
    {value, unit, cores, kind, sample} -- and the GPU agreeing with the CPU restatement on the whole workload."""
    env = {k: v for k, v in os.environ.items() if k not in ("WORLD_SIZE", "RANK", "LOCAL_RANK")}
    cmd = [sys.executable, os.path.join(ROOT, "bench.py"), "--steps", "4", "--warmup", "1", "--log2n", "14", "--no-secondary", "--no-host-resident",
           "--strong-total-log2n", "16"]
    env["SNARKV_BENCH_DETAILS"] = os.path.join(str(tmp_path), "details.json")
    r = subprocess.run(cmd, capture_output=True, text=True, timeout=900, cwd=ROOT, env=env)
    assert r.returncode == 0, r.stderr[-3000:]
    d, full = bench_line.parse(r.stdout)  # d = the LINE ITSELF: < 4 KB, last of stdout -- everything below is read off it
    assert "stages_ms" in full and "stages_ms" not in d  # the tables live in the record the line names
    assert d["config"]["single_msm_latency_ms"] > 0 and d["config"]["single_msm_points_per_s"] > 0 and "D2H" in d["config"]["timed_region"]
    for k in ("metric", "value", "unit", "n_gpus", "steps", "warmup", "ms_per_step", "higher_is_better", "scaling", "vs_baseline", "dtype",
              "data", "config", "roofline", "cpu_baseline"):
        assert k in d, k
    assert d["n_gpus"] == 1 and d["steps"] == 4 and d["warmup"] == 1 and d["higher_is_better"] is True and d["vs_baseline"] is None
    assert d["unit"] == "points/s" and d["scaling"] == "weak" and d["data"] == "synthetic" and "model" not in d["config"]
    assert "workload" in d["config"] and abs(d["value"] - (1 << 14) * 4 / (d["ms_per_step"] * 4e-3)) < 1e-6 * d["value"]
    rf = d["roofline"]
    assert rf["bound"] == "hbm" and rf["unit"] == "GB/s" and rf["peak"] == 8000.0 and abs(rf["frac"] - rf["achieved"] / rf["peak"]) < 1e-12
    assert "traffic" in rf and rf["achieved"] * 1e9 * rf["kernel_ms"] * 1e-3 == pytest.approx(96 * (1 << 14), rel=1e-9)
    cb = d["cpu_baseline"]
    assert all(k in cb for k in ("value", "unit", "cores", "kind", "sample")) and cb["kind"] == "port"
    assert cb["gpu_matches_on_sample"] is True and cb["sample_is_the_whole_workload"] is True
    assert d["config4_strong"]["matches_one_gpu_single_call"] is True and d["single_process_mgpu"]["job0_matches_one_gpu_recompute"] is True


def test_single_process_mgpu_leg_with_eight_emulated_ranks():
    """`snarkv_mgpu_*` with 8 ranks on the box's one device (what `bench.py --gpus 8` falls back to when no launcher
    works, and the form a Rust caller reaches without one): RCCL refuses eight ranks on one device loudly, the peer-copy
    transport delivers, job 0 equals the one-GPU recompute and the oracle."""
    import json

    env = {k: v for k, v in os.environ.items() if k not in ("WORLD_SIZE", "RANK", "LOCAL_RANK")}
    cmd = [sys.executable, os.path.join(ROOT, "bench.py"), "--mgpu-leg", "--gpus", "8", "--steps", "3", "--log2n", "12"]
    r = subprocess.run(cmd, capture_output=True, text=True, timeout=600, cwd=ROOT, env=env)
    assert r.returncode == 0, r.stderr[-3000:]
    mg = json.loads([ln for ln in r.stdout.splitlines() if ln.startswith("{")][-1])
    assert mg["ranks"] == 8 and mg["devices"] == [0] * 8 and "error" in mg["rccl"] and mg["peer_copy"]["value"] > 0
    assert mg["job0_matches_one_gpu_recompute"] is True and mg["results_distinct_per_input_set"] is True
    n = 8 << 12
    assert mg["result_job0"] == C.msm_pippenger(C.sample_scalars(0x5EED0001, n), C.sample_points(0x5EED0002, n), 8).hex()
