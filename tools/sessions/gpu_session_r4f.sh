#!/bin/bash
# Round-4 GPU session F: work list with alternating counters (no atomics in the fix-up), halo-2 windows of the centred kinds
set -u
REPO="${GRAFT_REPO_ROOT:-/root/repo}"
cd "$REPO"
O=gpurun_out/r4f; mkdir -p $O
export TMPDIR=/tmp
timeout 900 python -m pytest tests/test_gpu_parity.py tests/test_gpu_graph.py -q -p no:cacheprovider -x -k "advection or mac_cormack or baseline or tiled or graph" > $O/pytest_a.log 2>&1; echo "pytest parity rc=$?"; tail -3 $O/pytest_a.log
timeout 300 python bench.py --workload smoke256 --steps 20 --warmup 30 > $O/bench_smoke256.json 2> $O/bench_smoke256.err; echo "smoke256 rc=$?"
python - <<PY
import json
d=json.load(open('$O/bench_smoke256.json')); print(d['ms_per_step'], d['op_ms_profiled_step'], d['advect_fallback_last_call'], d['non_cg_share_of_profiled_step'])
PY
K=advect_self,mac_cormack_self,advect_centered,mac_cormack_centered
: > $O/time_frow.jsonl
for ROUND in 1 2; do
  timeout 300 python tools/time_frow.py --size 256 --only $K --lib phiflow_amd/lib/libphihip_r3.so >> $O/time_frow.jsonl 2>> $O/time_frow.err
  timeout 300 python tools/time_frow.py --size 256 --only $K >> $O/time_frow.jsonl 2>> $O/time_frow.err
  timeout 300 python tools/time_frow.py --size 256 --only $K --halo 2 >> $O/time_frow.jsonl 2>> $O/time_frow.err
done
for CFL in 1.3 1.8; do
  timeout 300 python tools/time_frow.py --size 256 --only $K --cfl $CFL --lib phiflow_amd/lib/libphihip_r3.so >> $O/time_frow.jsonl 2>> $O/time_frow.err
  timeout 300 python tools/time_frow.py --size 256 --only $K --cfl $CFL >> $O/time_frow.jsonl 2>> $O/time_frow.err
  timeout 300 python tools/time_frow.py --size 256 --only $K --cfl $CFL --halo 2 >> $O/time_frow.jsonl 2>> $O/time_frow.err
done
timeout 300 python tools/time_frow.py --size 256 --bc closed --only $K >> $O/time_frow.jsonl 2>> $O/time_frow.err
timeout 300 python tools/time_frow.py --size 256 --bc closed --only $K --halo 2 >> $O/time_frow.jsonl 2>> $O/time_frow.err
python - <<PY
import json
for l in open('$O/time_frow.jsonl'):
    d=json.loads(l)
    print(d['lib'][:16].ljust(16), d['size'], d['dtype'], d['bc'], 'cfl', d.get('cfl'), 'halo', d.get('halo'), ' '.join(f"{k}={v.get('ms','ERR')}" for k,v in d['kernels'].items()), d.get('advect_fallback'))
PY
