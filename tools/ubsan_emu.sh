#!/bin/bash
# UndefinedBehaviorSanitizer run of the kernel sources under the CPU emulation (-fno-sanitize-recover: the first report aborts the test).
# Usage: bash tools/ubsan_emu.sh [fuzz cases, default 30]
set -e
REPO="$(cd "$(dirname "$0")/.." && pwd)"
cd "$REPO"
SANITIZE=undefined bash tests/hipemu/build_emu.sh
export UBSAN_OPTIONS=print_stacktrace=1 LD_PRELOAD="$(gcc -print-file-name=libubsan.so)" PHIHIP_EMU_LIB="$REPO/tests/hipemu/libphihip_emu_ubsan.so"
N="${1:-30}"; T=$(( (N + 2) / 3 ))
for K in 0 1 2; do (python tests/fuzz_parity.py --emu --first $((K * T)) --count $T 2>&1 | grep -E "runtime error|ERROR|FAIL|fails|SUMMARY" || true) & done; wait
python -m pytest tests/test_emu_kernels.py tests/test_golden_emu.py -x -q -p no:cacheprovider 2>&1 | tail -25
