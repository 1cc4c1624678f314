#!/bin/bash
# Round-3 GPU session C: size scans (fp32 / fp64, tuned plans with the new tiles), bench, configs, gpu suite subset.
set -u
REPO="${GRAFT_REPO_ROOT:-/root/repo}"
cd "$REPO"
O=gpurun_out/r3c; mkdir -p $O
export TMPDIR=/tmp
STEPS="${1:-scan,bench,configs,test}"
if [[ "$STEPS" == *scan* ]]; then
  timeout 900 python tools/size_scan.py --sizes 192,224,256,288,320,384,448,512,640 --iters 30 > $O/size_scan.jsonl 2> $O/scan.err; echo "scan rc=$?"
  timeout 600 python tools/size_scan.py --sizes 256,320,384 --iters 30 --dtype f64 > $O/size_scan_f64.jsonl 2>> $O/scan.err; echo "scan64 rc=$?"
  python - <<PY
import json
for f in ('$O/size_scan.jsonl','$O/size_scan_f64.jsonl'):
    for l in open(f):
        d=json.loads(l); t=d.get('tuned') or d['model']
        print(d['size'], d['dtype'], t['plan_mv'][:3], t['plan_x2'][:3], t['plan_ur'][:3], t['us_matvec'], t['us_update_x2'], t['us_update_r'], t['us_iteration'], t['moved_GBs_matvec'], t['moved_GBs_iteration'])
PY
fi
if [[ "$STEPS" == *bench* ]]; then timeout 600 python bench.py --steps 20 --warmup 5 --pmc 0 --cpu-size 0 > $O/bench_quick.json 2> $O/bench.err; echo "bench rc=$?"; python -c "
import json; d=json.load(open('$O/bench_quick.json')); print(d['value'], d['ms_per_step'], d['roofline']['frac'], d['config3']['ms_per_iteration'], d['plan'], d['config3']['plan'])"; fi
if [[ "$STEPS" == *configs* ]]; then timeout 600 python tools/bench_configs.py 5 4 3 > $O/configs.jsonl 2> $O/configs.err; echo "configs rc=$?"; cut -c1-700 $O/configs.jsonl; fi
if [[ "$STEPS" == *test* ]]; then timeout 1800 python -m pytest tests -m gpu -q -p no:cacheprovider -s > $O/pytest_gpu.log 2>&1; echo "pytest rc=$?"; tail -4 $O/pytest_gpu.log; grep -n "closed 512\|config5 parity at\|config2 parity\|config3 parity" $O/pytest_gpu.log | cut -c1-400 | head; fi
