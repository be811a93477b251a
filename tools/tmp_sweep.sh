cd $GRAFT_REPO_ROOT
python -m pytest tests/test_gpu_msm.py tests/test_gpu_fullsize.py -x -q 2>&1 | tail -2
for r in 1 2 3; do
for acc in 1 0; do for inf in 3 4; do
  SNARKV_ACC_STREAM=$acc python bench.py --steps 40 --warmup 4 --no-cpu-baseline --no-secondary --inflight $inf 2>/dev/null | python -c "
import sys,json
d=json.loads(sys.stdin.read())
print('accstream=$acc inflight=$inf ms=%.4f acc_launch=%.3f frac=%.5f lat=%.3f' % (d['ms_per_step'], d['stages_ms']['bucket_accumulate'], d['roofline']['frac'], d['config']['single_msm_latency_ms']))"
done; done; done
SNARKV_ACC_STREAM=1 python tools/bench_large_msm.py 24 --lanes 3
SNARKV_ACC_STREAM=0 python tools/bench_large_msm.py 24 --lanes 3
