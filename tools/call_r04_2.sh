set -u
O=gpurun_out/c2; mkdir -p $O
python -m pytest tests -x -q -m gpu 2>&1 | tail -15 > $O/all_tests.txt; cat $O/all_tests.txt
python bench.py --steps 20 --warmup 5 > $O/bench.json 2> $O/bench.err; tail -3 $O/bench.err
python - <<'P'
import json
d=json.load(open("gpurun_out/c2/bench.json"))
print("value", d["value"], d["ms_per_step"], d["config"]["single_msm_latency_ms"])
print("roofline", {k:v for k,v in d["roofline"].items() if not isinstance(v,str)})
print("host_resident", json.dumps(d.get("host_resident"))[:1500])
print("named", json.dumps(d.get("named_configs")))
print("stages_seq", d.get("stages_ms_sequential"))
P
export TMPDIR=/tmp
python tools/pmc_traffic.py --tag r04 --out-dir $O > $O/pmc.log 2>&1; tail -25 $O/pmc.log
cd /tmp
for mode in "batch:--steps 20 --warmup 5" "sequential:--steps 20 --warmup 5 --inflight 1"; do
  name=${mode%%:*}; args=${mode#*:}
  rm -rf /tmp/prof_$name
  rocprofv3 --kernel-trace --stats -d /tmp/prof_$name -- python $GRAFT_REPO_ROOT/bench.py $args --no-cpu-baseline --no-secondary --no-host-resident > /dev/null 2>&1
  db=$(find /tmp/prof_$name -name "*.db" | head -1)
  if [ -n "$db" ]; then python $GRAFT_REPO_ROOT/tools/rocpd_top_kernels.py $db $GRAFT_REPO_ROOT/$O/r04_rocprofv3_kernel_stats_$name.csv; else
    f=$(find /tmp/prof_$name -name "*kernel_stats.csv" | head -1); [ -n "$f" ] && cp $f $GRAFT_REPO_ROOT/$O/r04_rocprofv3_kernel_stats_$name.csv; fi
done
cd $GRAFT_REPO_ROOT
head -12 $O/r04_rocprofv3_kernel_stats_sequential.csv
