#!/bin/bash
# Round-4 GPU session E: per-plane fix-up work list (advect_tile / advect_win): parity, the smoke256 bench line again, timings with and
# without lookups beyond the windows
set -u
REPO="${GRAFT_REPO_ROOT:-/root/repo}"
cd "$REPO"
O=gpurun_out/r4e; mkdir -p $O
export TMPDIR=/tmp
timeout 900 python -m pytest tests/test_gpu_parity.py -q -p no:cacheprovider -x -k "advection or mac_cormack or baseline or tiled" > $O/pytest_a.log 2>&1; echo "pytest parity rc=$?"; tail -3 $O/pytest_a.log
timeout 300 python bench.py --workload smoke256 --steps 20 --warmup 30 > $O/bench_smoke256.json 2> $O/bench_smoke256.err; echo "smoke256 rc=$?"
python - <<PY
import json
d=json.load(open('$O/bench_smoke256.json')); print(d['ms_per_step'], d['op_ms_profiled_step'], d['advect_fallback_last_call'], d['non_cg_share_of_profiled_step'])
PY
K=advect_self,mac_cormack_self,advect_centered,mac_cormack_centered
: > $O/time_frow.jsonl
for LIB in phiflow_amd/lib/libphihip_r3.so ""; do
  timeout 300 python tools/time_frow.py --size 256 --dtype f32 --bc periodic --only $K ${LIB:+--lib $LIB} >> $O/time_frow.jsonl 2>> $O/time_frow.err
  timeout 300 python tools/time_frow.py --size 256 --dtype f32 --bc periodic --only $K --cfl 1.3 ${LIB:+--lib $LIB} >> $O/time_frow.jsonl 2>> $O/time_frow.err
  timeout 300 python tools/time_frow.py --size 256 --dtype f32 --bc closed --only $K --cfl 1.3 ${LIB:+--lib $LIB} >> $O/time_frow.jsonl 2>> $O/time_frow.err
done
python - <<PY
import json
for l in open('$O/time_frow.jsonl'):
    d=json.loads(l)
    print(d['lib'][:20].ljust(20), d['size'], d['rank'], d['dtype'], d['bc'], d.get('cfl'), ' '.join(f"{k}={v.get('ms','ERR')}" for k,v in d['kernels'].items()), d.get('advect_fallback'))
PY
