set -u
O=gpurun_out/c3; mkdir -p $O
python -m pytest tests -x -q -m gpu 2>&1 | tail -15 > $O/all_tests.txt; cat $O/all_tests.txt
SNARKV_NAIVE_CHUNKS=1 SNARKV_NAIVE_JOINT=2 python -m pytest tests/test_gpu_robustness.py tests/test_gpu_msm.py tests/test_gpu_host_mirror.py tests/test_gpu_config5.py tests/test_gpu_plonk.py -x -q -m gpu -k "not fixed_window and not group_kernel and not chunk_base" 2>&1 | tail -4 > $O/forced_group.txt; cat $O/forced_group.txt
python bench.py --steps 20 --warmup 5 > $O/bench.json 2> $O/bench.err; tail -3 $O/bench.err
python - <<'P'
import json
d=json.load(open("gpurun_out/c3/bench.json"))
print("value", d["value"], d["ms_per_step"], d["config"]["single_msm_latency_ms"])
print("roofline", {k:v for k,v in d["roofline"].items() if not isinstance(v,str)})
print("host_resident", json.dumps(d.get("host_resident"))[:1800])
print("named", json.dumps(d.get("named_configs")))
P
for j in 1 2 3 4 5; do
  SNARKV_NAIVE_JOINT=$j python bench.py --steps 2 --warmup 1 --no-cpu-baseline --no-host-resident 2>/dev/null | python -c "
import json,sys; d=json.loads(sys.stdin.read()); s=d['secondary']
print('joint=$j', {k:round(s[k].get('ms') or s[k].get('ms_per_job'),4) for k in ('aggregate_64_proofs','aggregate_64_proofs_pipelined','aggregate_64_proofs_merged','aggregate_1024_proofs','aggregate_1024_proofs_pipelined','aggregate_1024_proofs_merged')})" | tee -a $O/ab_joint.txt
done
