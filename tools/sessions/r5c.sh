#!/bin/bash
# Round-5 GPU session C: the instruction diet of the LDS-staged advection kernels (sign-based tap addressing, v_fract, one v_max3 per sample, FULL-tile
# stores, no SLP packing) -- GPU parity of every advection test + randomised cases, then same-box A/B against the round-4 library (two alternating rounds)
set -u
REPO="${GRAFT_REPO_ROOT:-/root/repo}"; cd "$REPO"
O=gpurun_out/${SESSION_TAG:-r5c}; mkdir -p $O; export TMPDIR=/tmp
python -c "from phiflow_amd import _capi as C; l=C.load_default_library(); print('build', l.build_id(), 'tree', l.built_from_tree())" > $O/build_id.txt 2>&1; cat $O/build_id.txt
timeout 900 python -m pytest tests/test_gpu_parity.py tests/test_gpu_api.py tests/test_gpu_graph.py -m gpu -q -p no:cacheprovider -k "advect or mac or smoke or plume or step or graph or scene or fields or adjoint or obstacle or cellflags" > $O/pytest_adv.log 2>&1; echo "adv rc=$?"; tail -3 $O/pytest_adv.log
timeout 600 python tests/fuzz_parity.py --first 52000 --count 40 > $O/fuzz.log 2>&1; tail -1 $O/fuzz.log; grep "^FAIL" $O/fuzz.log | head -5
for ROUND in 1 2; do for LIB in phiflow_amd/lib/libphihip_r4.so ""; do
  timeout 300 python tools/time_frow.py --size 256 --dtype f32 --bc periodic ${LIB:+--lib $LIB} >> $O/time_frow.jsonl 2>> $O/time_frow.err
  timeout 300 python tools/time_frow.py --size 256 --dtype f32 --bc closed ${LIB:+--lib $LIB} >> $O/time_frow.jsonl 2>> $O/time_frow.err
  timeout 300 python tools/time_frow.py --size 384 --dtype f64 --bc closed ${LIB:+--lib $LIB} >> $O/time_frow.jsonl 2>> $O/time_frow.err
done; done
for LIB in phiflow_amd/lib/libphihip_r4.so ""; do
  timeout 300 python tools/time_frow.py --size 512 --dtype f32 --bc periodic --reps 10 ${LIB:+--lib $LIB} >> $O/time_frow.jsonl 2>> $O/time_frow.err
done
python - <<PY
import json
for l in open('$O/time_frow.jsonl'):
    d=json.loads(l)
    print(d['lib'][:16].ljust(16), d['size'], d['dtype'], d['bc'], ' '.join(f"{k}={v.get('ms','ERR')}" for k,v in d['kernels'].items()), d.get('advect_fallback'))
PY
