cd $GRAFT_REPO_ROOT
python -m pytest tests/test_gpu_msm.py -x -q -k "chunk or 2p24 or golden" 2>&1 | tail -2
for L in 22 24; do
  python bench.py --log2n $L --steps 6 --warmup 2 --no-cpu-baseline --no-secondary --inflight 1 2>/dev/null | python -c "
import sys,json
d=json.loads(sys.stdin.read()); sq=d.get('stages_ms_sequential') or {}
print('log2n=$L value=%.3e ms=%.3f lat=%.3f launch_n=%d' % (d['value'], d['ms_per_step'], d['config']['single_msm_latency_ms'], d['config']['points_per_kernel_launch']))
print('   stages', {k: round(v,3) for k,v in d['stages_ms'].items()}); print('   roofline', d['roofline']['achieved'], d['roofline']['frac'], d['valu_roofline']['frac'])"
done
