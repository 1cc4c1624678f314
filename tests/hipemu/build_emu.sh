#!/bin/bash
# TEST INFRASTRUCTURE ONLY: compiles the kernel sources of phiflow_amd/csrc against the fiber emulation with g++.
#   build_emu.sh            -> libphihip_emu.so
#   SANITIZE=1 build_emu.sh -> libphihip_emu_asan.so (AddressSanitizer: out-of-bounds accesses of the kernels on "device" buffers,
#                              LDS arrays and workspaces; run with LD_PRELOAD=$(gcc -print-file-name=libasan.so), see tools/asan_emu.sh)
#   SANITIZE=undefined build_emu.sh -> libphihip_emu_ubsan.so (LD_PRELOAD=$(gcc -print-file-name=libubsan.so))
set -e
here="$(cd "$(dirname "$0")" && pwd)"
src="$here/../../phiflow_amd/csrc"
if [ "${SANITIZE:-0}" = "undefined" ]; then
  out="$here/libphihip_emu_ubsan.so"; build="$here/build/ubsan"; extra="-fsanitize=undefined -fno-sanitize-recover=undefined"
elif [ "${SANITIZE:-0}" = "1" ]; then
  out="$here/libphihip_emu_asan.so"; build="$here/build/asan"; extra="-fsanitize=address -fno-omit-frame-pointer"
else
  out="$here/libphihip_emu.so"; build="$here/build"; extra=""
fi
mkdir -p "$build"
CXX="g++ -O1 -g -std=c++17 -fPIC -I$here/include -w $extra"
pids=()
$CXX -c "$here/hipemu.cpp" -o "$build/hipemu.o" & pids+=($!)
src_hash=$(cd "$src" && cat $(ls *.hip *.hpp | LC_ALL=C sort) ../../include/phihip.h | sha1sum | cut -c1-16)
$CXX -x c++ -DPHIHIP_BUILD_ID="\"emulation src:$src_hash\"" -c "$src/capi.hip" -o "$build/capi.o" & pids+=($!)
for f in cg project advect advect_tile advect_win adjoint cg_small cg_resident; do
  $CXX -x c++ -c "$src/$f.hip" -o "$build/$f.o" & pids+=($!)
done
for t in 0 1; do for d in 0 1; do
  $CXX -x c++ -DPHIHIP_INST_F64=$t -DPHIHIP_INST_DIM3=$d -c "$src/march_inst.hip" -o "$build/march_${t}_${d}.o" & pids+=($!)
done; done
for p in "${pids[@]}"; do wait $p; done
g++ -shared $extra -o "$out" "$build"/*.o
echo "built $out"
