#!/bin/bash
# Round-4 GPU session C: advect_win.hip after the grouped LDS layout + prepared constants; A/B: round-3 library, default build, variant with
# both tile positions unrolled
set -u
REPO="${GRAFT_REPO_ROOT:-/root/repo}"
cd "$REPO"
O=gpurun_out/r4c; mkdir -p $O
export TMPDIR=/tmp
timeout 900 python -m pytest tests/test_gpu_parity.py -q -p no:cacheprovider -x -k "advection or mac_cormack or baseline" > $O/pytest_a.log 2>&1; echo "pytest parity rc=$?"; tail -3 $O/pytest_a.log
: > $O/time_frow.jsonl
K=advect_self,mac_cormack_self,advect_centered,mac_cormack_centered
for ROUND in 1 2; do
  for LIB in phiflow_amd/lib/libphihip_r3.so "" phiflow_amd/lib/libphihip_unrollS.so; do
    timeout 300 python tools/time_frow.py --size 256 --dtype f32 --bc periodic --only $K ${LIB:+--lib $LIB} >> $O/time_frow.jsonl 2>> $O/time_frow.err
    timeout 300 python tools/time_frow.py --size 256 --dtype f32 --bc closed --only $K ${LIB:+--lib $LIB} >> $O/time_frow.jsonl 2>> $O/time_frow.err
  done
done
for LIB in phiflow_amd/lib/libphihip_r3.so "" phiflow_amd/lib/libphihip_unrollS.so; do
  timeout 300 python tools/time_frow.py --size 384 --dtype f64 --bc closed --only $K ${LIB:+--lib $LIB} >> $O/time_frow.jsonl 2>> $O/time_frow.err
  timeout 300 python tools/time_frow.py --size 256 --dtype f64 --bc periodic --only $K ${LIB:+--lib $LIB} >> $O/time_frow.jsonl 2>> $O/time_frow.err
  timeout 300 python tools/time_frow.py --size 512 --rank 2 --batch 8 --dtype f32 --bc closed --only $K ${LIB:+--lib $LIB} >> $O/time_frow.jsonl 2>> $O/time_frow.err
  timeout 300 python tools/time_frow.py --size 2048 --rank 2 --batch 1 --dtype f32 --bc closed --only $K ${LIB:+--lib $LIB} >> $O/time_frow.jsonl 2>> $O/time_frow.err
done
python - <<PY
import json
for l in open('$O/time_frow.jsonl'):
    d=json.loads(l)
    print(d['lib'][:20].ljust(20), d['size'], d['rank'], d['dtype'], d['bc'], ' '.join(f"{k}={v.get('ms','ERR')}" for k,v in d['kernels'].items()), d.get('advect_fallback'))
PY
