cd $GRAFT_REPO_ROOT
cat > /tmp/b.py <<'PY'
import os, sys, time
sys.path.insert(0, os.environ["GRAFT_REPO_ROOT"])
import torch, snark_verifier_amd as sv
k=int(sys.argv[1]); n=1<<k
ctx=sv.Context(0)
ds=torch.empty(32*n,dtype=torch.uint8,device="cuda"); dp=torch.empty(64*n,dtype=torch.uint8,device="cuda"); out=torch.zeros(64,dtype=torch.uint8,device="cuda")
torch.cuda.synchronize(); ctx.sample_scalars_dev(1,n,ds.data_ptr()); ctx.sample_points_dev(2,n,dp.data_ptr()); ctx.sync()
def f():
    ctx.msm_pippenger_dev(ds.data_ptr(),dp.data_ptr(),n,out.data_ptr(),0); ctx.sync()
f(); f(); t=time.perf_counter()
for _ in range(5): f()
print("2^%d workers=%s chunk=2^%s: %.2f ms"%(k,os.environ.get("SNARKV_SPLIT_WORKERS","3"),os.environ.get("SNARKV_SPLIT_LOG2","20"),(time.perf_counter()-t)/5*1e3))
PY
for k in 22 24; do for w in 2 3; do for cl in 19 20 21; do SNARKV_PIP_SPLIT=2 SNARKV_SPLIT_WORKERS=$w SNARKV_SPLIT_LOG2=$cl python /tmp/b.py $k; done; done; done
