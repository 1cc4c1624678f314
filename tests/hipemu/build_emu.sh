#!/bin/bash
# TEST INFRASTRUCTURE ONLY: compiles the kernel sources of phiflow_amd/csrc against the fiber emulation with g++.
set -e
here="$(cd "$(dirname "$0")" && pwd)"
src="$here/../../phiflow_amd/csrc"
out="$here/libphihip_emu.so"
build="$here/build"
mkdir -p "$build"
CXX="g++ -O1 -g -std=c++17 -fPIC -I$here/include -w"
pids=()
$CXX -c "$here/hipemu.cpp" -o "$build/hipemu.o" & pids+=($!)
for f in capi cg project advect adjoint cg_small; do
  $CXX -x c++ -c "$src/$f.hip" -o "$build/$f.o" & pids+=($!)
done
for t in 0 1; do for d in 0 1; do
  $CXX -x c++ -DPHIHIP_INST_F64=$t -DPHIHIP_INST_DIM3=$d -c "$src/march_inst.hip" -o "$build/march_${t}_${d}.o" & pids+=($!)
done; done
for p in "${pids[@]}"; do wait $p; done
g++ -shared -o "$out" "$build"/*.o
echo "built $out"
