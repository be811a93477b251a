set -u
O=gpurun_out/c5; mkdir -p $O
timeout 400 python -m pytest tests/test_gpu_poseidon.py tests/test_gpu_decider.py tests/test_gpu_plonk.py tests/test_gpu_config5.py -x -q -m gpu 2>&1 | tail -5 > $O/tests.txt; cat $O/tests.txt
timeout 300 python bench.py --steps 10 --warmup 3 --no-cpu-baseline --no-host-resident > $O/bench_quick.json 2> $O/bench_quick.err; tail -2 $O/bench_quick.err
python - <<'P'
import json
d=json.load(open("gpurun_out/c5/bench_quick.json"))
print("value", d["value"], d["ms_per_step"])
s=d["secondary"]
for k,v in s.items():
    if isinstance(v,dict): print(k, v.get("ms") or v.get("ms_per_job"))
P
