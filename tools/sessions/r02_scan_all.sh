#!/bin/bash
# size scans (model vs first-call autotune) fp32 + fp64 and the bench line
REPO="${GRAFT_REPO_ROOT:-/root/repo}"; cd "$REPO"; mkdir -p gpurun_out; export TMPDIR=/tmp
TAG=${1:-b}
timeout 400 python tools/size_scan.py --sizes 192,224,256,288,320,384,448,512,640 > gpurun_out/r02_size_scan_$TAG.jsonl 2> gpurun_out/r02_size_scan_$TAG.err; echo "scan rc=$?"
timeout 300 python tools/size_scan.py --sizes 256,384 --dtype f64 > gpurun_out/r02_size_scan_f64_$TAG.jsonl 2>> gpurun_out/r02_size_scan_$TAG.err; echo "scan64 rc=$?"
timeout 400 python bench.py --steps 20 --warmup 5 --pmc 0 > gpurun_out/r02_bench_$TAG.json 2> gpurun_out/r02_bench_$TAG.err; echo "bench rc=$?"
python - $TAG <<'PY'
import json,sys
t=sys.argv[1]
for f in (f'gpurun_out/r02_size_scan_{t}.jsonl', f'gpurun_out/r02_size_scan_f64_{t}.jsonl'):
    for l in open(f):
        d=json.loads(l); print(d['size'], d['dtype'], {k:(v['us_matvec'],v['us_update_x2'],v['us_update_r'],v['us_iteration'],v['moved_GBs_iteration'],v['plan_mv'][:4],v['plan_x2'][:4],v['plan_ur'][:4]) for k,v in d.items() if isinstance(v,dict)})
d=json.loads(open(f'gpurun_out/r02_bench_{t}.json').read())
print(d['value'], d['ms_per_step'], d['kernel_ms_per_step'], d['config3']['ms_per_iteration'], d['config3']['moved_frac'], d['config3']['plan'])
PY
