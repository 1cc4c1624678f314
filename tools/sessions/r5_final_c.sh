#!/bin/bash
# Round-5 final session, third part (same build as r5_final.sh; after tools/path_workload.py learnt the name of the LDS-DMA kernel of the fp64 group and
# bench.py's traffic cross-check learnt to compare launch plans): the driver-style bench line and the per-kernel roofline table once more
set -u
REPO="${GRAFT_REPO_ROOT:-/root/repo}"; cd "$REPO"
O=gpurun_out/${SESSION_TAG:-r5z}; mkdir -p $O; export TMPDIR=/tmp
python -c "from phiflow_amd import _capi as C; l=C.load_default_library(); print('build', l.build_id(), 'tree', l.built_from_tree())" > $O/build_id_c.txt 2>&1; cat $O/build_id_c.txt
timeout 900 python bench.py --steps 20 --warmup 5 > $O/bench_n1.json 2> $O/bench_n1.err; echo "bench rc=$?"; head -c 300 $O/bench_n1.json; echo
rm -rf $O/roofline; timeout 1200 bash tools/kernel_roofline.sh $O/roofline > $O/roofline.log 2>&1; tail -3 $O/roofline.log
find $O -name "*kernel_trace.csv" -size +2M -delete 2>/dev/null; find $O -name "*counter_collection.csv" -size +2M -delete 2>/dev/null
python - <<PY
import json
d=json.loads(open('$O/bench_n1.json').read().strip().splitlines()[-1])
print('ms/step', d['ms_per_step'], 'roofline', {k: d['roofline'].get(k) for k in ('frac','achieved','traffic_over_moved','traffic_cross_check')}, 'config3', d['config3']['ms_per_iteration'], d['config3']['plan']['matvec'])
t=json.load(open('$O/roofline/kernel_roofline.json'))
for g in t['groups']:
    print(g['group'], g.get('cg_iteration_check',{}).get('ratio'))
    for k in g['kernels']:
        if k.get('launches',0) and ('a1' in k['label'] or 'CG' in k['label']): print('  ', k['label'][:70], k['avg_us'], k.get('frac_of_8TBs'), k.get('pmc_over_moved'))
PY
echo finished
