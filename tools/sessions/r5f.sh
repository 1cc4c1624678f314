#!/bin/bash
# Round-5 GPU session F: fence placement of the register-staged tile kernel (closed boxes: the grids the LDS-DMA kernel cannot take), bench line with the
# PMC child passes pinned to the invocation's own launch plans (traffic cross-check), smoke256 line
set -u
REPO="${GRAFT_REPO_ROOT:-/root/repo}"; cd "$REPO"
O=gpurun_out/${SESSION_TAG:-r5f}; mkdir -p $O; export TMPDIR=/tmp
python -c "from phiflow_amd import _capi as C; l=C.load_default_library(); print('build', l.build_id(), 'tree', l.built_from_tree())" > $O/build_id.txt 2>&1; cat $O/build_id.txt
for ROUND in 1 2; do for LIB in "" phiflow_amd/lib/libphihip_t2.so phiflow_amd/lib/libphihip_t0.so; do
  timeout 300 python tools/time_frow.py --size 256 --dtype f32 --bc closed --only advect_self,mac_cormack_self --reps 40 ${LIB:+--lib $LIB} >> $O/time_fence.jsonl 2>> $O/time_fence.err
  timeout 300 python tools/time_frow.py --size 384 --dtype f64 --bc closed --only advect_self --reps 10 ${LIB:+--lib $LIB} >> $O/time_fence.jsonl 2>> $O/time_fence.err
  PHIHIP_ADVECT_DMA=0 timeout 300 python tools/time_frow.py --size 256 --dtype f32 --bc periodic --only advect_self --reps 40 ${LIB:+--lib $LIB} >> $O/time_fence.jsonl 2>> $O/time_fence.err
  timeout 300 python tools/time_frow.py --size 1024 --rank 2 --batch 8 --dtype f32 --bc closed --only advect_self --reps 40 ${LIB:+--lib $LIB} >> $O/time_fence.jsonl 2>> $O/time_fence.err
done; done
python - <<PY
import json
for l in open('$O/time_fence.jsonl'):
    d=json.loads(l)
    print(d['lib'][:18].ljust(18), d['size'], d['rank'], d['dtype'], d['bc'], ' '.join(f"{k}={v.get('ms','ERR')}" for k,v in d['kernels'].items()))
PY
timeout 900 python bench.py --steps 20 --warmup 5 --phi-level 0 --cpu-size 0 > $O/bench_n1.json 2> $O/bench_n1.err; echo "bench rc=$?"; python - <<PY
import json
d=json.load(open('$O/bench_n1.json'))
print('ms/step', d['ms_per_step'], 'roofline', {k:d['roofline'].get(k) for k in ('frac','achieved','traffic','traffic_over_moved','traffic_cross_check')})
print(d['roofline']['traffic_source'])
PY
timeout 300 python bench.py --workload smoke256 --steps 20 --warmup 30 > $O/bench_smoke256.json 2> $O/bench_smoke256.err; echo "smoke256 rc=$?"; python - <<PY
import json
d=json.load(open('$O/bench_smoke256.json')); print('smoke256 ms/step', d['ms_per_step']); print({k:v for k,v in d.items() if 'kernel' in k or 'share' in k})
PY
