"""Shared by the bench.py tests: parse the ONE contract line (the last `{` line of stdout), check what the driver needs of
it -- it is the LAST non-empty line, under 4 KB (round 5's 26.5 KB line was cut by the driver's log tail and the round went
unmeasured) -- and return (compact line, full record read from the `details` file the line names)."""
import json
import os

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
LINE_LIMIT = 4096
CONTRACT_KEYS = ("metric", "value", "unit", "n_gpus", "steps", "warmup", "ms_per_step", "higher_is_better", "scaling", "vs_baseline", "dtype",
                 "data", "config", "details")


def parse(stdout):
    lines = [ln for ln in stdout.splitlines() if ln.strip()]
    json_lines = [ln for ln in lines if ln.startswith("{")]
    assert len(json_lines) == 1, stdout[-2000:]  # ONE line, from rank 0 only
    assert lines[-1] is json_lines[0] or lines[-1] == json_lines[0], "the contract line must be the LAST line of stdout"
    assert len(json_lines[0]) < LINE_LIMIT, len(json_lines[0])
    c = json.loads(json_lines[0])
    for k in CONTRACT_KEYS:
        assert k in c, k
    assert "workload" in c["config"] and "model" not in c["config"]
    path = c["details"] if os.path.isabs(c["details"]) else os.path.join(ROOT, c["details"])
    with open(path) as f:
        full = json.load(f)
    for k in ("metric", "value", "unit", "n_gpus", "steps", "warmup", "ms_per_step", "higher_is_better", "scaling", "vs_baseline", "data"):
        assert full[k] == c[k], k  # the line is an extract of the record, not a second measurement
    return c, full
