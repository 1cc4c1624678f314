#!/bin/bash
# Round-4 GPU session H (re-entry after the container was re-created): is HEAD green on the GPU, the driver-style bench line, the smoke256
# line and a same-box A/B of the f-row kernels against the round-3 library
set -u
REPO="${GRAFT_REPO_ROOT:-/root/repo}"
cd "$REPO"
O=gpurun_out/r4h; mkdir -p $O
export TMPDIR=/tmp
python -c "from phiflow_amd import _capi as C; l=C.load_default_library(); print('build', l.build_id(), 'tree', l.built_from_tree())" > $O/build_id.txt 2>&1; cat $O/build_id.txt
timeout 900 python -m pytest tests -m gpu -q -p no:cacheprovider > $O/pytest_gpu.log 2>&1; echo "pytest rc=$?"; tail -2 $O/pytest_gpu.log
timeout 300 python -c "import __graft_entry__ as g; g.smoke()" > $O/smoke.log 2>&1; echo "smoke rc=$?"; tail -1 $O/smoke.log
timeout 600 python bench.py --steps 20 --warmup 5 > $O/bench_n1.json 2> $O/bench_n1.err; echo "bench rc=$?"; head -c 400 $O/bench_n1.json; echo
timeout 300 python bench.py --workload smoke256 --steps 20 --warmup 30 > $O/bench_smoke256.json 2> $O/bench_smoke256.err; echo "smoke256 rc=$?"; head -c 300 $O/bench_smoke256.json; echo
: > $O/time_frow.jsonl
for ROUND in 1 2; do
  for LIB in phiflow_amd/lib/libphihip_r3.so ""; do
    timeout 300 python tools/time_frow.py --size 256 --dtype f32 --bc periodic ${LIB:+--lib $LIB} >> $O/time_frow.jsonl 2>> $O/time_frow.err
    timeout 300 python tools/time_frow.py --size 256 --dtype f32 --bc closed ${LIB:+--lib $LIB} >> $O/time_frow.jsonl 2>> $O/time_frow.err
    timeout 300 python tools/time_frow.py --size 384 --dtype f64 --bc closed ${LIB:+--lib $LIB} >> $O/time_frow.jsonl 2>> $O/time_frow.err
  done
done
python - <<PY
import json
for l in open('$O/time_frow.jsonl'):
    d=json.loads(l)
    print(d['lib'][:16].ljust(16), d['size'], d['dtype'], d['bc'], 'cfl', d.get('cfl'), ' '.join(f"{k}={v.get('ms','ERR')}" for k,v in d['kernels'].items()), d.get('advect_fallback'))
PY
