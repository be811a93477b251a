cd $GRAFT_REPO_ROOT
for r in 1 2 3; do
for inf in 2 3 4; do for st in 20 40; do
  python bench.py --steps $st --warmup 4 --no-cpu-baseline --no-secondary --inflight $inf 2>/dev/null | python -c "
import sys,json
d=json.loads(sys.stdin.read())
print('inflight=$inf steps=$st ms=%.4f acc_launch=%.3f frac=%.5f' % (d['ms_per_step'], d['stages_ms']['bucket_accumulate'], d['roofline']['frac']))"
done; done; done
