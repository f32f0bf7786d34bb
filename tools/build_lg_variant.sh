#!/bin/bash
# ab/<name>.so = the product objects with k_levelgen.hip rebuilt under extra -D flags (the seed / level-generation kernels only; a flag that
# changes a struct shared with other objects needs tools/build_variant.sh).   tools/build_lg_variant.sh nobar -DMGX_EXP_NO_SW_BARRIER
set -e
name=$1; shift
cd "$(dirname "$0")/../gym-minigrid_amd/csrc"
mkdir -p ../../ab
make -j8 libmgx.so > /dev/null
/opt/rocm/bin/hipcc --offload-arch=gfx950 -O3 -std=c++17 -fPIC -I../../include -I. "$@" -c -o /tmp/k_levelgen_$name.o k_levelgen.hip
/opt/rocm/bin/hipcc --offload-arch=gfx950 -shared -o ../../ab/$name.so k_step.o /tmp/k_levelgen_$name.o k_epilogue.o k_dynobs.o k_state.o mgx_api.o levelgen.o
echo "built ab/$name.so ($*)"
