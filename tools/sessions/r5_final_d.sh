#!/bin/bash
# Round-5 final session, last part (same library build as r5_final.sh; after the Python-level changes that followed it -- jit_compile's bookkeeping, the
# residual-based parity check of tolerance solves): the GPU suite, smoke and the driver-style bench line once more, and the two torch-level records
set -u
REPO="${GRAFT_REPO_ROOT:-/root/repo}"; cd "$REPO"
O=gpurun_out/${SESSION_TAG:-r5z}; mkdir -p $O; export TMPDIR=/tmp
python -c "from phiflow_amd import _capi as C; l=C.load_default_library(); print('build', l.build_id(), 'tree', l.built_from_tree())" > $O/build_id.txt 2>&1; cat $O/build_id.txt
timeout 1500 python -m pytest tests -m gpu -q -p no:cacheprovider > $O/pytest_gpu.log 2>&1; echo "pytest rc=$?"; tail -2 $O/pytest_gpu.log
timeout 300 python -c "import __graft_entry__ as g; g.smoke()" > $O/smoke.log 2>&1; echo "smoke rc=$?"; tail -1 $O/smoke.log
timeout 900 python bench.py --steps 20 --warmup 5 > $O/bench_n1.json 2> $O/bench_n1.err; echo "bench rc=$?"; head -c 300 $O/bench_n1.json; echo
timeout 120 python tools/micro/foreach_copy_check.py > $O/foreach_copy_check.txt 2>&1
timeout 200 python tools/micro/jit_foreach_debug.py > $O/jit_foreach_debug.txt 2>&1; grep "first difference" $O/jit_foreach_debug.txt
timeout 120 python tools/micro/nan_fill_check.py -1 128 40 > $O/nan_fill_check.txt 2>&1; tail -2 $O/nan_fill_check.txt
echo finished
