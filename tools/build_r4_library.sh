#!/bin/bash
# Rebuilds the round-4 library (commit f679723, the state the round-4 verdict judged) as phiflow_amd/lib/libphihip_r4.so -- the A/B partner of
# tools/sessions/r5_final.sh (same box, alternating rounds). It is NOT kept in the tree between sessions: a second library in phiflow_amd/lib/ travels to
# every GPU box (housekeeping). ~3 min of hipcc.          bash tools/build_r4_library.sh
set -e
cd "$(dirname "$0")/.."
W=$(mktemp -d /tmp/phihip_r4.XXXXXX)
git worktree add --detach "$W" f679723 > /dev/null
make -C "$W/phiflow_amd/csrc" -j8 OUT="$PWD/phiflow_amd/lib/libphihip_r4.so" > /dev/null
git worktree remove --force "$W"
ls -la phiflow_amd/lib/libphihip_r4.so
