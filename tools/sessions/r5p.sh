#!/bin/bash
# Round-5 GPU session P: three block orders of the sawtooth -- plain (no sawtooth), full mirror of the block order, XCD-preserving reversal (HEAD)
set -u
REPO="${GRAFT_REPO_ROOT:-/root/repo}"; cd "$REPO"
O=gpurun_out/${SESSION_TAG:-r5p}; mkdir -p $O; export TMPDIR=/tmp
LIBS="phiflow_amd/lib/libphihip_plain.so phiflow_amd/lib/libphihip_sawfull.so phiflow_amd/lib/libphihip.so"
CFG="1,64,32;1,32,32;1,64,16;2,64,64;1,64,64;4,64,64"
for N in 256 384 512; do for LIB in $LIBS; do
  timeout 300 python tools/sweep_cg.py --size $N --iters 40 --configs "$CFG" --lib $LIB >> $O/sweep.jsonl 2>> $O/sweep.err
done; done
python - <<PY
import json
rows=[json.loads(l) for l in open('$O/sweep.jsonl') if l.startswith('{')]
by={}
for r in rows: by.setdefault((r['size'], tuple(r['plan_mv'][:3]) if r['plan_mv'] else None, r['plan_mv'][3] if r['plan_mv'] else 0),{})[r['lib'][9:]]=r['ms_iter_wall']
for k,v in by.items(): print(k, v)
PY
for ROUND in 1 2; do for LIB in $LIBS; do
  PHIHIP_LIBRARY=$LIB timeout 600 python bench.py --steps 20 --warmup 5 --pmc 0 --cpu-size 0 --phi-level 0 > $O/bench_${ROUND}_$(basename $LIB .so).json 2>> $O/bench.err
  python -c "import json;d=json.loads(open('$O/bench_${ROUND}_$(basename $LIB .so).json').read().strip().splitlines()[-1]);print('bench', '$LIB', d['ms_per_step'], d['roofline']['frac'], d.get('config3',{}).get('ms_per_iteration'))"
done; done
for LIB in $LIBS; do
  timeout 300 python tools/size_scan.py --sizes 256,384 --iters 40 --dtype f64 --bc 1 --flags 1 --lib $LIB 2>> $O/scan.err | sed "s|^{|{\"lib\": \"$(basename $LIB .so)\", |" >> $O/scan_f64_flags.jsonl
done
python - <<PY
import json
for r in [json.loads(l) for l in open('$O/scan_f64_flags.jsonl') if l.startswith('{')]: print(r['lib'], r['size'], r['tuned']['us_iteration'])
PY
echo finished
