#!/bin/bash
# Round-5 final session, second part (same build, same gpurun call): 120 randomised parity cases (resident-solver arm on every 2-D case; its tolerance-mode checks
# are limited to grids a white-noise right-hand side converges on within 1000 iterations) and the fp64 roofline group once more with the advection's chunk pinned
set -u
REPO="${GRAFT_REPO_ROOT:-/root/repo}"; cd "$REPO"
O=gpurun_out/${SESSION_TAG:-r5z}; mkdir -p $O; export TMPDIR=/tmp
python -c "from phiflow_amd import _capi as C; l=C.load_default_library(); print('build', l.build_id(), 'tree', l.built_from_tree())" > $O/build_id_b.txt 2>&1; cat $O/build_id_b.txt
timeout 900 python tests/fuzz_parity.py --first 53000 --count 120 > $O/fuzz.log 2>&1; tail -1 $O/fuzz.log; grep "^FAIL" $O/fuzz.log | head -5
timeout 900 bash tools/kernel_roofline.sh $O/roofline_f64 f64_384 > $O/roofline_f64.log 2>&1; tail -2 $O/roofline_f64.log
python - <<PY
import json
d=json.load(open('$O/roofline_f64/kernel_roofline.json'))
for g in d['groups']:
    print(g['group'], g.get('cg_iteration_check'))
    for k in g['kernels']:
        if k.get('launches',0): print('  ', k['label'][:60], k['avg_us'], k['frac_of_8TBs'], k.get('pmc_over_moved'))
PY
find $O -name "*kernel_trace.csv" -size +2M -delete 2>/dev/null; find $O -name "*counter_collection.csv" -size +2M -delete 2>/dev/null
