#!/bin/bash
# Runs ON THE GPU BOX (via gpurun): every record the round's profiles/ directory keeps, written under gpurun_out/<tag>_*
# (gpurun merges only gpurun_out/ back; copy what you want judged into profiles/).
#   tools/collect_profiles.sh r06
set -u
TAG=${1:-r06}
R=$GRAFT_REPO_ROOT; O=$R/gpurun_out; mkdir -p $O
export TMPDIR=/tmp
cd $R
# PMC records first (and into profiles/ of THIS tree) so that the bench lines below quote the traffic of these very kernels
timeout 600 python tools/pmc_traffic.py --tag $TAG --out-dir $O > /dev/null
timeout 600 python tools/pmc_traffic.py --tag $TAG --out-dir $O --log2n 24 > /dev/null
cp $O/${TAG}_pmc_hbm_traffic*.json profiles/
# the static ISA census of k_accumulate on these sources (needs only hipcc): issue_roofline of the bench line
timeout 600 python tools/isa_stats.py --tag $TAG > $O/${TAG}_isa_stats.txt 2>&1; cp profiles/${TAG}_isa_k_accumulate.json $O/ 2>/dev/null
cd /tmp
export SNARKV_BENCH_DETAILS=/tmp/bench_details_profiled.json  # (profiled runs: their records are not the round's figures)
stats() {  # name, bench args...
  name=$1; shift
  rm -rf /tmp/prof_$name
  timeout 600 rocprofv3 --kernel-trace --stats -d /tmp/prof_$name -- python $R/bench.py "$@" > /dev/null 2>&1
  db=$(find /tmp/prof_$name -name "*.db" | head -1)
  if [ -n "$db" ]; then python $R/tools/rocpd_top_kernels.py $db $O/${TAG}_rocprofv3_kernel_stats_$name.csv; else
    f=$(find /tmp/prof_$name -name "*kernel_stats.csv" | head -1); [ -n "$f" ] && cp $f $O/${TAG}_rocprofv3_kernel_stats_$name.csv; fi
}
stats batch --steps 20 --warmup 5 --no-cpu-baseline --no-secondary --no-host-resident --no-strong --no-mgpu-leg
stats sequential --steps 20 --warmup 5 --no-cpu-baseline --no-secondary --inflight 1 --no-strong --no-mgpu-leg
# the hash sidecar bench.py reads `kernel_us_profile` from (keyed to THESE sources), into profiles/ of this tree
cp $O/${TAG}_rocprofv3_kernel_stats_sequential.csv $R/profiles/ && python $R/tools/stats_sidecar.py $R/profiles/${TAG}_rocprofv3_kernel_stats_sequential.csv \
  && cp $R/profiles/${TAG}_kernel_stats_sequential.json $O/
stats with_secondary --steps 4 --warmup 1 --no-cpu-baseline --no-host-resident --no-strong --no-mgpu-leg
stats 2p24_single --log2n 24 --inflight 1 --steps 4 --warmup 1 --no-cpu-baseline --no-secondary --no-strong --no-mgpu-leg
cd $R
# (the line bench.py prints is the compact contract line; the full record goes to the file SNARKV_BENCH_DETAILS names)
SNARKV_BENCH_DETAILS=$O/${TAG}_bench_details.json timeout 600 python bench.py > $O/${TAG}_bench.json 2> $O/${TAG}_bench.err
export SNARKV_BENCH_DETAILS=$O/${TAG}_bench_details_other.json
timeout 300 python bench.py --steps 40 --warmup 4 --no-secondary --no-cpu-baseline --no-strong --no-mgpu-leg > $O/${TAG}_bench_40steps.json 2>/dev/null
timeout 300 python bench.py --log2n 22 --inflight 1 --steps 10 --warmup 2 --no-secondary --cpu-sample-log2 18 --no-strong --no-mgpu-leg > $O/${TAG}_bench_2p22.json 2>/dev/null
timeout 300 python bench.py --log2n 24 --inflight 1 --steps 6 --warmup 2 --no-secondary --cpu-sample-log2 18 --no-strong --no-mgpu-leg > $O/${TAG}_bench_2p24.json 2>/dev/null
# every keyed record the main line quotes must be of THESE sources: fail loudly otherwise
TAGV=$TAG python - <<'PY' || { echo "STALE PROFILE RECORDS: see above" | tee $O/${TAG}_STALE.txt; }
import json, os, sys
c = json.loads([l for l in open("gpurun_out/%s_bench.json" % os.environ["TAGV"]) if l.startswith("{")][-1])
assert len(json.dumps(c)) < 4096
d = json.load(open("gpurun_out/%s_bench_details.json" % os.environ["TAGV"]))
bad = []
if d.get("issue_roofline", {}).get("frac") is None: bad.append("issue_roofline: " + str(d.get("issue_roofline", {}).get("source")))
if d["roofline"].get("traffic") is None: bad.append("roofline.traffic: " + str(d["roofline"].get("traffic_source")))
if d["roofline"].get("kernel_us_profile") is None: bad.append("roofline.kernel_us_profile: " + str(d["roofline"].get("kernel_us_profile_source")))
print("\n".join(bad) or "keyed records all match the kernel sources")
sys.exit(1 if bad else 0)
PY
cd /tmp
# the timed batch as a kernel timeline (who overlaps whom): the window from the 45th k_prepare (2 x 20 initialisation / warm-up
# jobs + the 4 slot calls) to the end of the batch's k_final
rm -rf /tmp/prof_trace; timeout 600 rocprofv3 --kernel-trace --output-format csv -d /tmp/prof_trace -- python $R/bench.py --steps 20 --warmup 5 --no-cpu-baseline --no-secondary --no-host-resident --no-strong --no-mgpu-leg > /dev/null 2>&1
python $R/tools/trace_overlap.py /tmp/prof_trace --first 44 --count 20 --until-prepare 64 > $O/${TAG}_overlap_batch.txt 2>&1
# the aggregation job alone (bench.py's second metric): kernel shares for secondary.aggregate_*.roofline.dominant_kernel
for m in 64 1024; do
  rm -rf /tmp/prof_a; timeout 600 rocprofv3 --kernel-trace --stats -d /tmp/prof_a -- python $R/tools/aggregate_job.py --proofs $m > /dev/null 2>&1
  db=$(find /tmp/prof_a -name "*.db" | head -1)
  if [ -n "$db" ]; then python $R/tools/rocpd_top_kernels.py $db $O/${TAG}_rocprofv3_kernel_stats_aggregate_$m.csv; else
    f=$(find /tmp/prof_a -name "*kernel_stats.csv" | head -1); [ -n "$f" ] && cp $f $O/${TAG}_rocprofv3_kernel_stats_aggregate_$m.csv; fi
done
# ... and with jobs in flight: by hardware-queue count, and the kernel durations under 16-fold overlap
cd $R
GPU_MAX_HW_QUEUES=16 timeout 600 python tools/aggregate_inflight.py --inflight 1 8 16 32 2>/dev/null | grep queues= > $O/${TAG}_agg_inflight.txt
cd /tmp
for m in 64 1024; do
  rm -rf /tmp/prof_a; timeout 600 rocprofv3 --kernel-trace --stats -d /tmp/prof_a -- python $R/tools/aggregate_inflight.py --proofs $m --inflight 16 > /dev/null 2>&1
  db=$(find /tmp/prof_a -name "*.db" | head -1)
  if [ -n "$db" ]; then python $R/tools/rocpd_top_kernels.py $db $O/${TAG}_rocprofv3_kernel_stats_aggregate_inflight16_$m.csv; else
    f=$(find /tmp/prof_a -name "*kernel_stats.csv" | head -1); [ -n "$f" ] && cp $f $O/${TAG}_rocprofv3_kernel_stats_aggregate_inflight16_$m.csv; fi
done
cd $R
timeout 120 python tools/dk_create_time.py 2>/dev/null | grep -v amdgpu.ids > $O/${TAG}_dk_create.txt
timeout 300 python tools/ab_decide.py --sizes 1,16,256,1024 2>/dev/null | grep -v amdgpu.ids > $O/${TAG}_ab_decide_now.txt
ls -la $O | grep $TAG
