#!/bin/bash
# Round-4 GPU session S: LDS-window advection passes with 4-row tiles (one position per thread: ~139 instead of ~230 VGPRs, 43 instead of 71 KB of LDS
# for the staggered MacCormack correction) against the 8-row tiles that ship -- same box, alternating
set -u
REPO="${GRAFT_REPO_ROOT:-/root/repo}"
cd "$REPO"
O=gpurun_out/r4s; mkdir -p $O
export TMPDIR=/tmp
: > $O/time_frow_t1.jsonl
for ROUND in 1 2; do
  for LIB in phiflow_amd/lib/libphihip_exp.so ""; do
    timeout 300 python tools/time_frow.py --size 256 --dtype f32 --bc periodic ${LIB:+--lib $LIB} >> $O/time_frow_t1.jsonl 2>> $O/time_frow_t1.err
    timeout 300 python tools/time_frow.py --size 256 --dtype f32 --bc closed ${LIB:+--lib $LIB} >> $O/time_frow_t1.jsonl 2>> $O/time_frow_t1.err
  done
done
for LIB in phiflow_amd/lib/libphihip_exp.so ""; do
  timeout 300 python tools/time_frow.py --size 512 --dtype f32 --bc periodic --reps 10 ${LIB:+--lib $LIB} >> $O/time_frow_t1.jsonl 2>> $O/time_frow_t1.err
done
python - <<PY
import json
for l in open('$O/time_frow_t1.jsonl'):
    d=json.loads(l)
    print(d['lib'][:16].ljust(16), d['size'], d['dtype'], d['bc'], ' '.join(f"{k}={v.get('ms','ERR')}" for k,v in d['kernels'].items() if 'advect' in k or 'mac' in k), d.get('advect_fallback'))
PY
