#!/bin/bash
# dev tool: rebuild ONE unit (csrc/<unit>.hip) with extra flags and link it with the current objects of the other units
# into tools/tmp/libsnarkv_<name>.so (the default build is left untouched)
#   tools/build_variant_unit.sh msm_naive bitserial "-DSNARKV_NAIVE_WINDOW=0"
set -e
cd "$(dirname "$0")/.."
mkdir -p tools/tmp
B=snark-verifier_amd/build
/opt/rocm/bin/hipcc --offload-arch=gfx950 -O3 -std=c++17 -fPIC -fno-gpu-rdc -Wno-unused-result $3 -c snark-verifier_amd/csrc/$1.hip -o tools/tmp/$1_$2.o
OBJS=""
for u in capi msm_naive msm_pippenger decider sample poseidon ipa mgpu decompress; do if [ $u != $1 ]; then OBJS="$OBJS $B/$u.o"; fi; done
/opt/rocm/bin/hipcc --offload-arch=gfx950 -shared -fPIC -o tools/tmp/libsnarkv_$2.so $OBJS tools/tmp/$1_$2.o
echo built tools/tmp/libsnarkv_$2.so
