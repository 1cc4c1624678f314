#!/bin/bash
# serialised build of libphihip.so (+ optionally the CPU emulation library): bash tools/build.sh [emu]
set -e
REPO="$(cd "$(dirname "$0")/.." && pwd)"
flock "$REPO/phiflow_amd/csrc/.build.lock" make -C "$REPO/phiflow_amd/csrc" -j"$(nproc)" ARCH=gfx950 | tail -1
if [ "${1:-}" = "emu" ]; then flock "$REPO/phiflow_amd/csrc/.build.lock" bash "$REPO/tests/hipemu/build_emu.sh" | tail -1; fi
