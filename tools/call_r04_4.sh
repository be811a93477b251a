set -u
O=gpurun_out/c4; mkdir -p $O
python -m pytest tests -x -q -m gpu 2>&1 | grep -E "passed|failed|error" | tail -3 > $O/all_tests.txt; cat $O/all_tests.txt
SNARKV_NAIVE_CHUNKS=1 SNARKV_NAIVE_JOINT=1 python -m pytest tests/test_gpu_robustness.py tests/test_gpu_msm.py tests/test_gpu_host_mirror.py tests/test_gpu_config5.py tests/test_gpu_plonk.py -x -q -m gpu -k "not fixed_window and not group_kernel and not chunk_base" 2>&1 | grep -E "passed|failed|error" | tail -3 > $O/forced_group.txt; cat $O/forced_group.txt
python tools/ab_decide.py --sizes 1,16,256,1024 > $O/ab_decide.txt 2>&1; cat $O/ab_decide.txt
python - > $O/dk_create.txt 2>&1 <<'P'
import time, sys, os
sys.path.insert(0, os.getcwd())
import snark_verifier_amd as sv
ctx = sv.Context(0)
g2 = bytes.fromhex("edf692d95cbdde46ddda5ef7d422436779445c5e66006a42761e1f12efde0018c212f3aeb785e49712e7a9353349aaf1255dfb31b7bf60723a480d9293938e19aa7dfa6601cce64c7bd3430c69e7d1e38f40cb8d8071ab4aeb6d8cdba55ec8125b9722d1dcdaac55f38eb37033314bbc95330c69ad999eec75f05f58d0890609")
g1 = (1).to_bytes(32, "little") + (2).to_bytes(32, "little")
for _ in range(3): sv.DecidingKey(ctx, g1, g2, g2).close()
t0 = time.perf_counter()
for _ in range(20): sv.DecidingKey(ctx, g1, g2, g2).close()
print("snarkv_dk_create (line tables of g2, -s_g2 on one wavefront): %.3f ms per key" % ((time.perf_counter() - t0) / 20 * 1e3))
lib = sv.load_library()
acc = g1 + g1
for _ in range(3): lib.bn254_kzg_decide(g1, g2, g2, acc)
t0 = time.perf_counter()
for _ in range(20): assert lib.bn254_kzg_decide(g1, g2, g2, acc) == 1
print("bn254_kzg_decide (context-free, same key: tables kept): %.3f ms per call" % ((time.perf_counter() - t0) / 20 * 1e3))
P
cat $O/dk_create.txt
python bench.py --steps 20 --warmup 5 > $O/bench.json 2> $O/bench.err; tail -3 $O/bench.err
python - <<'P'
import json
d=json.load(open("gpurun_out/c4/bench.json"))
print("value", d["value"], d["ms_per_step"], d["config"]["single_msm_latency_ms"])
print("host_resident", d["host_resident"]["value"], d["host_resident"]["fraction_of_pcie_bound"], d["host_resident"]["in_memory_form"]["value"])
print("named", json.dumps(d.get("named_configs")))
s=d["secondary"]
print({k:(v.get("ms") or v.get("ms_per_job")) for k,v in s.items() if isinstance(v,dict)})
P
