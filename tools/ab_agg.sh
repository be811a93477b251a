#!/bin/bash
# dev tool (GPU box): interleaved A/B of library variants on the aggregation job (tools/aggregate_job.py), medians
#   tools/ab_agg.sh <rounds> <proofs> default bitserial ...
cd "$(dirname "$0")/.."
ROUNDS=$1; PROOFS=$2; shift; shift
for r in $(seq 1 $ROUNDS); do
  for v in "$@"; do
    if [ "$v" != default ]; then export SNARKV_AMD_LIB=$PWD/tools/tmp/libsnarkv_$v.so; else unset SNARKV_AMD_LIB; fi
    python tools/aggregate_job.py --proofs $PROOFS --reps 30 2>/dev/null | sed "s/^/AB $v /"
  done
done | tee /tmp/ab_agg.txt
python - <<'PY'
import re,statistics,collections
d=collections.defaultdict(list)
for l in open('/tmp/ab_agg.txt'):
    m=re.match(r"AB (\S+) .*: ([0-9.]+) ms",l)
    if m: d[m.group(1)].append(float(m.group(2)))
for v,x in d.items(): print("MEDIAN %-10s ms_per_job=%.4f (n=%d)"%(v,statistics.median(x),len(x)))
PY
