#!/bin/bash
# dev tool (GPU box): interleaved A/B of ENVIRONMENT variants of one build (knobs the launchers read per call), medians.
#   tools/ab_env.sh <rounds> "NAME=VAR=VAL[,VAR=VAL]" ...        e.g.  tools/ab_env.sh 5 "off=SNARKV_PAIR_TREE=0" "on=SNARKV_PAIR_TREE=1"
cd "$(dirname "$0")/.."
ROUNDS=$1; shift
for r in $(seq 1 $ROUNDS); do
  for spec in "$@"; do
    name=${spec%%=*}; vars=${spec#*=}
    env $(echo $vars | tr ',' ' ') python bench.py --steps ${STEPS:-40} --warmup 4 --no-cpu-baseline --no-secondary ${EXTRA:-} 2>/dev/null | python -c "
import sys,json
c=json.loads(sys.stdin.read().strip().splitlines()[-1]); d=json.load(open(c["details"])); st=d.get('stages_ms') or {}
print('AB %-10s ms=%.4f lat=%.4f acc_in_batch=%.4f comb_in_batch=%.4f' % ('$name', d['ms_per_step'], d['config']['single_msm_latency_ms'] or 0, st.get('bucket_accumulate',0), st.get('bucket_combine',0)))"
  done
done | tee /tmp/ab_env.txt
python - <<'PY'
import re,statistics,collections
d=collections.defaultdict(lambda: collections.defaultdict(list))
for l in open('/tmp/ab_env.txt'):
    m=re.match(r"AB (\S+)\s+(.*)",l)
    if not m: continue
    for kv in m.group(2).split():
        k,v=kv.split('='); d[m.group(1)][k].append(float(v))
for v,kv in d.items():
    print("MEDIAN %-10s "%v+" ".join("%s=%.4f"%(k,statistics.median(x)) for k,x in kv.items()))
PY
