#!/bin/bash
# dev tool: build libsnarkv_amd.so with extra flags into tools/tmp/<name>.so, then restore the default build
#   tools/build_variant.sh prof "-DSNARKV_DECIDE_PROFILE"
set -e
cd "$(dirname "$0")/.."
mkdir -p tools/tmp
touch snark-verifier_amd/csrc/*.hip
SNARKV_EXTRA_FLAGS="$2" python -c "import sys; sys.path.insert(0,'snark-verifier_amd'); import build; build.build()"
cp snark-verifier_amd/libsnarkv_amd.so tools/tmp/libsnarkv_$1.so
touch snark-verifier_amd/csrc/*.hip
python -c "import sys; sys.path.insert(0,'snark-verifier_amd'); import build; build.build()"
