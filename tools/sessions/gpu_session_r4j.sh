#!/bin/bash
# Round-4 GPU session J: the resident 2-D solver with the tagged-granule exchange -- parity on the GPU, sweep against the launch forms
set -u
REPO="${GRAFT_REPO_ROOT:-/root/repo}"
cd "$REPO"
O=gpurun_out/r4j; mkdir -p $O
export TMPDIR=/tmp
python -c "from phiflow_amd import _capi as C; l=C.load_default_library(); print('build', l.build_id(), 'tree', l.built_from_tree())" > $O/build_id.txt 2>&1; cat $O/build_id.txt
timeout 600 python -m pytest tests/test_gpu_parity.py -m gpu -q -p no:cacheprovider -k "resident" -x -s > $O/pytest_resident.log 2>&1; echo "pytest rc=$?"; tail -4 $O/pytest_resident.log
timeout 600 python tools/sweep_resident.py 400 > $O/sweep_resident.jsonl 2> $O/sweep_resident.err; echo "sweep rc=$?"; tail -2 $O/sweep_resident.err
python - <<PY
import json
for l in open('$O/sweep_resident.jsonl'):
    d=json.loads(l)
    print(d['res'], d['batch'], d['bc'], 'launches', d['launches']['us_per_iteration'], 'resident', d['resident']['us_per_iteration'], 'x', d['speedup_resident'],
          'tol ms', d['launches']['tolerance_solve']['ms'], d['resident']['tolerance_solve']['ms'], d['launches']['tolerance_solve']['iterations'][:2], d['resident']['tolerance_solve']['iterations'][:2], 'relL2 %.1e' % d['rel_l2_resident_vs_launches'])
PY
