#!/bin/bash
# dev tool (GPU box): interleaved A/B of library variants -- thermal / DVFS drift between consecutive runs on a box is
# larger than most kernel changes, so variants alternate and medians are compared.
#   tools/ab.sh <rounds> default base krun64 ...      (names = tools/tmp/libsnarkv_<name>.so; "default" = the tree's)
cd "$(dirname "$0")/.."
ROUNDS=$1; shift
for r in $(seq 1 $ROUNDS); do
  for v in "$@"; do
    if [ "$v" != default ]; then export SNARKV_AMD_LIB=$PWD/tools/tmp/libsnarkv_$v.so; else unset SNARKV_AMD_LIB; fi
    python bench.py --steps 40 --warmup 4 --no-cpu-baseline --no-secondary --inflight ${INFLIGHT:-0} ${EXTRA:-} 2>/dev/null | python -c "
import sys,json
c=json.loads(sys.stdin.read().strip().splitlines()[-1]); d=json.load(open(c['details'])); sq=d.get('stages_ms_sequential') or {}  # the line names the full record
print('AB %-10s ms=%.4f lat=%.4f acc=%.4f comb=%.4f red=%.4f prep=%.4f sort=%.4f shift=%.4f' % ('$v', d['ms_per_step'], d['config']['single_msm_latency_ms'] or 0, sq.get('bucket_accumulate',0), sq.get('bucket_combine',0), sq.get('bucket_reduce',0), sq.get('prepare_glv_montgomery_histogram',0), sq.get('partition_sort',0), sq.get('window_shift_chain',0)))"
  done
done | tee /tmp/ab.txt
python - <<'PY'
import re,statistics,collections
d=collections.defaultdict(lambda: collections.defaultdict(list))
for l in open('/tmp/ab.txt'):
    m=re.match(r"AB (\S+)\s+(.*)",l)
    if not m: continue
    for kv in m.group(2).split():
        k,v=kv.split('='); d[m.group(1)][k].append(float(v))
for v,kv in d.items():
    print("MEDIAN %-10s "%v+" ".join("%s=%.4f"%(k,statistics.median(x)) for k,x in kv.items()))
PY
