#!/bin/bash
# Build libmgx.so with extra -D flags into ab/<name>.so (tuning aid for same-box A/B runs: tools/ab.sh).
#   tools/build_variant.sh base            tools/build_variant.sh prio -DMGX_EXP_PRIO=1
set -e
name=$1; shift
cd "$(dirname "$0")/../gym-minigrid_amd/csrc"
mkdir -p ../../ab
make clean >/dev/null   # every object: a flag may change a struct shared by host and kernels
make -j8 libmgx.so CXXFLAGS="-O3 -std=c++17 -fPIC -Wall -Wextra -Wno-unused-parameter $*" 2>&1 | grep -E "error|warning" || true
cp libmgx.so ../../ab/$name.so
echo "built ab/$name.so ($*)  -- NOTE: csrc/ now holds this variant's objects: run make clean && make there for the product build"
