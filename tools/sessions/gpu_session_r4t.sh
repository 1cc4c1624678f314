#!/bin/bash
# Round-4 GPU session T: randomised parity cases (tests/fuzz_parity.py) on the final build, three seed ranges incl. larger grids
set -u
REPO="${GRAFT_REPO_ROOT:-/root/repo}"
cd "$REPO"
O=gpurun_out/r4t; mkdir -p $O
export TMPDIR=/tmp
python -c "from phiflow_amd import _capi as C; l=C.load_default_library(); print('build', l.build_id(), 'tree', l.built_from_tree())" > $O/build_id.txt 2>&1; cat $O/build_id.txt
timeout 900 python tests/fuzz_parity.py --first 50000 --count 150 > $O/fuzz_a.log 2>&1; tail -1 $O/fuzz_a.log; grep "^FAIL" $O/fuzz_a.log | head -5
timeout 900 python tests/fuzz_parity.py --first 60000 --count 40 --max-res 72 > $O/fuzz_b.log 2>&1; tail -1 $O/fuzz_b.log; grep "^FAIL" $O/fuzz_b.log | head -5
