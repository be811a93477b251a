#!/bin/bash
# the host mirror with its stage traces compiled in (dev aid; the product library has none)
set -e
R=$(cd "$(dirname "$0")/.." && pwd)
g++ -O3 -mbmi2 -madx -std=c++17 -shared -fPIC -DSNARKV_HOST_TRACE=1 -o "$R/snark-verifier_amd/libsnarkv_host_trace.so" \
    "$R/snark-verifier_amd/host/capi.cpp" -L"$R/snark-verifier_amd" -lsnarkv_amd -pthread -Wl,-rpath,'$ORIGIN'
echo built
