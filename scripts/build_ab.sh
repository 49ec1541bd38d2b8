#!/bin/bash
# A/B build of ONE translation unit: scripts/build_ab.sh spconv_tiles.hip libefg_hip_ab.so -DEFG_TILE_EARLYB=0
# links the alternate object with the other objects of efg_amd/lib/ (build the main library first); run a leg with
# EFG_HIP_LIB_AB=libefg_hip_ab.so.  The alternate .so is git-ignored and travels to the GPU box like the main one.
set -e
src=$1; out=$2; shift 2
root=$(cd "$(dirname "$0")/.." && pwd)
lib=$root/efg_amd/lib
obj=$lib/ab_$(basename "${src%.*}").o
/opt/rocm/bin/hipcc --offload-arch=gfx950 -O3 -std=c++17 -fPIC -munsafe-fp-atomics -ffp-contract=off -Wno-unused-function \
  -I$root/include -I$root/efg_amd/csrc "$@" -x hip -c $root/efg_amd/csrc/$src -o $obj
others=$(ls $lib/*.o | grep -v "/ab_" | grep -v "/$(basename "${src%.*}").o")
/opt/rocm/bin/hipcc --offload-arch=gfx950 -shared -fPIC -o $lib/$out $obj $others
echo built $lib/$out
