export TMPDIR=/tmp; R=$GRAFT_REPO_ROOT; cd /tmp
rocprofv3 --kernel-trace --pmc SQ_INSTS_VALU SQ_INSTS_SALU SQ_WAVES SQ_ACTIVE_INST_VALU SQ_WAVE_CYCLES SQ_BUSY_CYCLES --output-format csv -d $R/gpurun_out/pmc_sq -- python $R/bench.py --steps 3 --warmup 1 --no-cpu-baseline --no-secondary --inflight 1 > /dev/null 2>$R/gpurun_out/pmc_sq.err
cd $R; ls gpurun_out/pmc_sq/*/ | head; python - <<'PY'
import csv,glob,re,collections
f=glob.glob('gpurun_out/pmc_sq/**/*counter_collection.csv',recursive=True)[0]
rows=list(csv.DictReader(open(f)))
print(rows[0].keys())
agg=collections.defaultdict(lambda: collections.defaultdict(float)); cnt=collections.Counter()
seen=set()
for r in rows:
    m=re.search(r"(k_[a-z0-9_]+)",r["Kernel_Name"]); n=m.group(1) if m else r["Kernel_Name"][:24]
    agg[n][r["Counter_Name"]]+=float(r["Counter_Value"])
    key=(n,r["Dispatch_Id"])
    if key not in seen: seen.add(key); cnt[n]+=1
names=sorted({c for v in agg.values() for c in v})
print("%-24s %5s "%("kernel","calls")+" ".join("%16s"%c for c in names))
for n,v in sorted(agg.items(), key=lambda kv:-kv[1].get("SQ_INSTS_VALU",0)):
    print("%-24s %5d "%(n,cnt[n])+" ".join("%16.4g"%(v.get(c,0)/cnt[n]) for c in names))
PY
