#!/bin/bash
# Round-5 GPU session N: sawtooth marching (UPDATE kernels walk their chunks back to front, -DPHIHIP_SAWTOOTH=1) against the plain order, two builds of the
# same sources, two alternating rounds: CG iteration by size (tools/size_scan.py), fp64 + flags, the benchmark line
set -u
REPO="${GRAFT_REPO_ROOT:-/root/repo}"; cd "$REPO"
O=gpurun_out/${SESSION_TAG:-r5n}; mkdir -p $O; export TMPDIR=/tmp
SAW=phiflow_amd/lib/libphihip.so
for ROUND in 1 2; do for LIB in phiflow_amd/lib/libphihip_plain.so $SAW; do
  timeout 300 python tools/size_scan.py --sizes 192,256,288,320,384,448,512 --iters 60 ${LIB:+--lib $LIB} 2>> $O/scan.err | sed "s|^{|{\"lib\": \"$(basename $LIB .so)\", |" >> $O/scan_f32.jsonl
  timeout 300 python tools/size_scan.py --sizes 256,384 --iters 40 --dtype f64 --bc 1 --flags 1 ${LIB:+--lib $LIB} 2>> $O/scan.err | sed "s|^{|{\"lib\": \"$(basename $LIB .so)\", |" >> $O/scan_f64_flags.jsonl
done; done
python - <<PY
import json
for f in ('scan_f32','scan_f64_flags'):
    rows=[json.loads(l) for l in open('$O/'+f+'.jsonl') if l.startswith('{')]
    keys=[k for k in rows[0] if 'iter' in k or 'us' in k or 'ms' in k][:3]
    for r in rows: print(f, r['lib'][-12:], r.get('n'), {k: r[k] for k in keys})
PY
for ROUND in 1 2; do for LIB in phiflow_amd/lib/libphihip_plain.so $SAW; do
  PHIHIP_LIBRARY=$LIB timeout 600 python bench.py --steps 20 --warmup 5 --pmc 0 --cpu-size 0 --phi-level 0 > $O/bench_${ROUND}_$(basename ${LIB:-plain} .so).json 2>> $O/bench.err
  python -c "import json;d=json.loads(open('$O/bench_${ROUND}_$(basename ${LIB:-plain} .so).json').read().strip().splitlines()[-1]);print('bench', '${LIB:-plain}', d['ms_per_step'], d['roofline']['frac'], d.get('config3',{}).get('ms_per_iteration'))"
done; done
echo finished
