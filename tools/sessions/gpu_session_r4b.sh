#!/bin/bash
# Round-4 GPU session B: the LDS-windowed MacCormack / centred advection kernels (advect_win.hip): parity on the GPU, then timings
# against the round-3 library on the same box
set -u
REPO="${GRAFT_REPO_ROOT:-/root/repo}"
cd "$REPO"
O=gpurun_out/r4b; mkdir -p $O
export TMPDIR=/tmp
timeout 900 python -m pytest tests/test_gpu_parity.py -q -p no:cacheprovider -x -k "advection or mac_cormack or baseline or stencils" > $O/pytest_a.log 2>&1; echo "pytest parity rc=$?"; tail -3 $O/pytest_a.log
: > $O/time_frow.jsonl
K=advect_self,mac_cormack_self,advect_centered,mac_cormack_centered
for ROUND in 1 2; do
  for LIB in phiflow_amd/lib/libphihip_r3.so ""; do
    timeout 300 python tools/time_frow.py --size 256 --dtype f32 --bc periodic --only $K ${LIB:+--lib $LIB} >> $O/time_frow.jsonl 2>> $O/time_frow.err
  done
done
for LIB in phiflow_amd/lib/libphihip_r3.so ""; do
  timeout 300 python tools/time_frow.py --size 384 --dtype f64 --bc closed --only $K ${LIB:+--lib $LIB} >> $O/time_frow.jsonl 2>> $O/time_frow.err
  timeout 300 python tools/time_frow.py --size 256 --dtype f32 --bc closed --only $K ${LIB:+--lib $LIB} >> $O/time_frow.jsonl 2>> $O/time_frow.err
  timeout 300 python tools/time_frow.py --size 512 --rank 2 --batch 8 --dtype f32 --bc closed --only $K ${LIB:+--lib $LIB} >> $O/time_frow.jsonl 2>> $O/time_frow.err
  timeout 300 python tools/time_frow.py --size 512 --dtype f32 --bc periodic --reps 10 --only $K ${LIB:+--lib $LIB} >> $O/time_frow.jsonl 2>> $O/time_frow.err
done
python - <<PY
import json
for l in open('$O/time_frow.jsonl'):
    d=json.loads(l)
    print(d['lib'][:16].ljust(16), d['size'], d['rank'], d['dtype'], d['bc'], ' '.join(f"{k}={v.get('ms','ERR')}" for k,v in d['kernels'].items()), d.get('advect_fallback'))
PY
cd /tmp && timeout 300 rocprofv3 --kernel-trace --stats --output-format csv -d $REPO/$O/prof -o k -- python $REPO/tools/time_frow.py --size 256 --only $K --reps 10 > $REPO/$O/prof.log 2>&1; echo "rocprof rc=$?"
cd $REPO; python - <<PY
import csv,glob
for f in glob.glob('$O/prof/**/*kernel_stats.csv', recursive=True):
    for r in list(csv.DictReader(open(f)))[:12]:
        print(r['Name'][:90], r['Calls'], r['AverageNs'])
PY
find $O/prof -name "*kernel_trace.csv" -size +1M -delete 2>/dev/null
