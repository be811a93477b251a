set -u
O=gpurun_out/c5; mkdir -p $O; rm -f $O/e2e_trace.txt
timeout 120 python -m pytest tests/test_gpu_poseidon.py -x -q -m gpu 2>&1 | tail -15 > $O/tests0.txt; cat $O/tests0.txt
timeout 300 python -m pytest tests/test_gpu_plonk.py tests/test_gpu_host_mirror.py tests/test_gpu_config5.py -x -q -m gpu 2>&1 | tail -15 > $O/tests.txt; cat $O/tests.txt
for kind in 2; do
  for rep in 16 1; do
    echo "== kind $kind rep $rep" >> $O/e2e_trace.txt
    SNARKV_HOST_LIB=$PWD/snark-verifier_amd/libsnarkv_host_trace.so timeout 120 python tools/e2e_trace.py --kind $kind --rep $rep --calls 6 >> $O/e2e_trace.txt 2>&1
  done
done
grep "read_proofs_device\|^call" $O/e2e_trace.txt
