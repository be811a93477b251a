cd $GRAFT_REPO_ROOT
python -m pytest tests/test_gpu_msm.py tests/test_gpu_fullsize.py -x -q 2>&1 | tail -3
for rep in 1 2 3; do
  python bench.py --steps 24 --warmup 4 --no-cpu-baseline --no-secondary --inflight 4 2>/dev/null | python -c "
import sys,json
d=json.loads(sys.stdin.read())
sq=d['stages_ms_sequential']
print('packed64 value=%.3e ms=%.3f lat=%.3f seq: acc=%.3f comb=%.3f red=%.3f prep=%.3f sort=%.3f shift=%.3f' % (d['value'], d['ms_per_step'], d['config']['single_msm_latency_ms'], sq['bucket_accumulate'], sq['bucket_combine'], sq['bucket_reduce'], sq['prepare_glv_montgomery_histogram'], sq['partition_sort'], sq['window_shift_chain']))"
done
python tools/pmc_traffic.py --tag r02 --out-dir gpurun_out | head -8
