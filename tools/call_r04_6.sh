set -u
O=gpurun_out/c6; mkdir -p $O; rm -f $O/*.txt
run() { # name, env...
  name=$1; shift
  env "$@" timeout 120 python tools/e2e_trace.py --kind 0 --rep 16 --calls 25 2>/dev/null | python -c "
import sys,re
fa=[];tot=[]
for l in sys.stdin:
    m=re.search(r\"'fr_algebra': ([0-9.]+).*'total': ([0-9.]+)\", l)
    if m: fa.append(float(m.group(1))); tot.append(float(m.group(2)))
fa=fa[2:]; tot=tot[2:]
print('$name', 'fr_algebra min %.2f med %.2f max %.2f | total min %.2f med %.2f max %.2f | outliers(>1.5x med) %d/%d' % (min(fa), sorted(fa)[len(fa)//2], max(fa), min(tot), sorted(tot)[len(tot)//2], max(tot), sum(1 for x in tot if x>1.5*sorted(tot)[len(tot)//2]), len(tot)))
" >> $O/outliers.txt
}
run base A=1
run trim MALLOC_TRIM_THRESHOLD_=4000000000 MALLOC_TOP_PAD_=268435456
run arena8 MALLOC_ARENA_MAX=8
run pool32 SNARKV_HOST_POOL=32
run pool16 SNARKV_HOST_POOL=16
run base2 A=1
cat $O/outliers.txt
