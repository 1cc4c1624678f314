#!/bin/bash
# AddressSanitizer run of the kernel sources under the CPU emulation (the reference has no sanitizer coverage; a GPU
# would not report an out-of-bounds halo or workspace access at all). Builds tests/hipemu/libphihip_emu_asan.so and drives it with
# the randomised parity cases and the emulation test-suite. Usage: bash tools/asan_emu.sh [fuzz cases, default 60]
set -e
REPO="$(cd "$(dirname "$0")/.." && pwd)"
cd "$REPO"
SANITIZE=1 bash tests/hipemu/build_emu.sh
export ASAN_OPTIONS=detect_leaks=0 LD_PRELOAD="$(gcc -print-file-name=libasan.so)" PHIHIP_EMU_LIB="$REPO/tests/hipemu/libphihip_emu_asan.so"
# the randomised cases in three processes (a case under the sanitizer takes minutes since the resident-solver arm exists), then the emulation suite on the host's cores
N="${1:-60}"; T=$(( (N + 2) / 3 ))
for K in 0 1 2; do (python tests/fuzz_parity.py --emu --first $((K * T)) --count $T 2>&1 | grep -E "ERROR|FAIL|fails|SUMMARY" || true) & done; wait
python -m pytest tests/test_emu_kernels.py tests/test_golden_emu.py -x -q -p no:cacheprovider 2>&1 | tail -25
