#!/bin/bash
# Round-5 GPU session E: fence placement / unrolling of the LDS-DMA advection kernel (throw-away variant libraries), the bench line with the new
# `phi_level` block, the differentiated step on HEAD
set -u
REPO="${GRAFT_REPO_ROOT:-/root/repo}"; cd "$REPO"
O=gpurun_out/${SESSION_TAG:-r5e}; mkdir -p $O; export TMPDIR=/tmp
python -c "from phiflow_amd import _capi as C; l=C.load_default_library(); print('build', l.build_id(), 'tree', l.built_from_tree())" > $O/build_id.txt 2>&1; cat $O/build_id.txt
for ROUND in 1 2; do for LIB in "" phiflow_amd/lib/libphihip_f2.so phiflow_amd/lib/libphihip_f0.so phiflow_amd/lib/libphihip_f0u.so phiflow_amd/lib/libphihip_f2u.so; do
  timeout 300 python tools/time_frow.py --size 256 --dtype f32 --bc periodic --only advect_self --reps 60 ${LIB:+--lib $LIB} >> $O/time_fence.jsonl 2>> $O/time_fence.err
  timeout 300 python tools/time_frow.py --size 512 --dtype f32 --bc periodic --only advect_self --reps 10 ${LIB:+--lib $LIB} >> $O/time_fence.jsonl 2>> $O/time_fence.err
  timeout 300 python tools/time_frow.py --size 256 --dtype f64 --bc periodic --only advect_self --reps 30 ${LIB:+--lib $LIB} >> $O/time_fence.jsonl 2>> $O/time_fence.err
done; done
python - <<PY
import json
for l in open('$O/time_fence.jsonl'):
    d=json.loads(l)
    print(d['lib'][:18].ljust(18), d['size'], d['dtype'], d['bc'], ' '.join(f"{k}={v.get('ms','ERR')}" for k,v in d['kernels'].items()))
PY
timeout 900 python bench.py --steps 20 --warmup 5 > $O/bench_n1.json 2> $O/bench_n1.err; echo "bench rc=$?"; python - <<PY
import json
d=json.load(open('$O/bench_n1.json'))
print('ms/step', d['ms_per_step'], 'roofline', {k:d['roofline'].get(k) for k in ('frac','achieved','traffic','traffic_over_moved')})
print('phi_level', json.dumps(d.get('phi_level')))
PY
tail -3 $O/bench_n1.err
timeout 600 python tools/time_backward_step.py > $O/backward_step.jsonl 2> $O/backward_step.err; echo "bwd rc=$?"; cat $O/backward_step.jsonl | cut -c1-600
