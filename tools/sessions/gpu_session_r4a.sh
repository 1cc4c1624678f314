#!/bin/bash
# Round-4 GPU session A: parity of the vector divergence / resample kernels and of diffuse.explicit on the marching kernels, then the
# non-CG kernels of a step timed against the round-3 library on the same box (tools/time_frow.py)
set -u
REPO="${GRAFT_REPO_ROOT:-/root/repo}"
cd "$REPO"
O=gpurun_out/r4a; mkdir -p $O
export TMPDIR=/tmp
timeout 900 python -m pytest tests/test_gpu_parity.py -q -p no:cacheprovider -x -k "stencils or mac_cormack_and_resample or wall_velocity or make_incompressible" > $O/pytest_a.log 2>&1; echo "pytest parity rc=$?"; tail -3 $O/pytest_a.log
timeout 600 python -m pytest tests/test_gpu_api.py -q -p no:cacheprovider -x -k "host_api" > $O/pytest_b.log 2>&1; echo "pytest api rc=$?"; tail -3 $O/pytest_b.log
: > $O/time_frow.jsonl
for ROUND in 1 2; do
  for LIB in phiflow_amd/lib/libphihip_r3.so ""; do
    timeout 300 python tools/time_frow.py --size 256 --dtype f32 --bc periodic ${LIB:+--lib $LIB} >> $O/time_frow.jsonl 2>> $O/time_frow.err
    timeout 300 python tools/time_frow.py --size 384 --dtype f64 --bc closed ${LIB:+--lib $LIB} >> $O/time_frow.jsonl 2>> $O/time_frow.err
  done
done
for LIB in phiflow_amd/lib/libphihip_r3.so ""; do
  timeout 300 python tools/time_frow.py --size 256 --dtype f32 --bc closed ${LIB:+--lib $LIB} >> $O/time_frow.jsonl 2>> $O/time_frow.err
  timeout 300 python tools/time_frow.py --size 512 --rank 2 --batch 8 --dtype f32 --bc closed ${LIB:+--lib $LIB} >> $O/time_frow.jsonl 2>> $O/time_frow.err
  timeout 300 python tools/time_frow.py --size 512 --dtype f32 --bc periodic --reps 10 ${LIB:+--lib $LIB} >> $O/time_frow.jsonl 2>> $O/time_frow.err
done
python - <<PY
import json
for l in open('$O/time_frow.jsonl'):
    d=json.loads(l)
    print(d['lib'][:16].ljust(16), d['size'], d['rank'], d['dtype'], d['bc'], ' '.join(f"{k}={v.get('ms','ERR')}" for k,v in d['kernels'].items()))
PY
