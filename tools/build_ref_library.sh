#!/bin/bash
# Builds the library of an EARLIER commit as phiflow_amd/lib/libphihip_<name>.so -- the same-box A/B partner of a GPU session
# (AB_LIBS=phiflow_amd/lib/libphihip_r5.so bash tools/gpu_session.sh TAG time_frow:256/f32/periodic ...). It is NOT kept in the tree between sessions: a second
# library in phiflow_amd/lib/ travels to every GPU box. ~4 min of hipcc.          bash tools/build_ref_library.sh fd9a4ee r5
set -e
cd "$(dirname "$0")/.."
COMMIT="${1:?commit}"; NAME="${2:?name}"
W=$(mktemp -d /tmp/phihip_ref.XXXXXX)
git worktree add --detach "$W" "$COMMIT" > /dev/null
make -C "$W/phiflow_amd/csrc" -j8 OUT="$PWD/phiflow_amd/lib/libphihip_$NAME.so" > /dev/null
git worktree remove --force "$W"
ls -la phiflow_amd/lib/libphihip_$NAME.so
