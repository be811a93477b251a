#!/bin/bash
# dev tool: rebuild ONLY csrc/msm_pippenger.hip with extra flags and link it with the current objects of the other
# units into tools/tmp/libsnarkv_<name>.so (the default build is left untouched)
#   tools/build_variant_fast.sh acc4 "-DSNARKV_ACC_WAVES=4"
set -e
cd "$(dirname "$0")/.."
mkdir -p tools/tmp
B=snark-verifier_amd/build
/opt/rocm/bin/hipcc --offload-arch=gfx950 -O3 -std=c++17 -fPIC -fno-gpu-rdc -Wno-unused-result $2 -c snark-verifier_amd/csrc/msm_pippenger.hip -o tools/tmp/msm_pippenger_$1.o
OBJS=""
for u in capi msm_naive decider sample poseidon ipa mgpu; do OBJS="$OBJS $B/$u.o"; done
/opt/rocm/bin/hipcc --offload-arch=gfx950 -shared -fPIC -o tools/tmp/libsnarkv_$1.so $OBJS tools/tmp/msm_pippenger_$1.o
echo built tools/tmp/libsnarkv_$1.so
