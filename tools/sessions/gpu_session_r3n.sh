#!/bin/bash
# Round-3 GPU session N (final build): smoke, the whole GPU suite, the bench line, per-kernel roofline table (three groups), configs 3-5,
# the config4 / slab lines, rocprofv3 stats of the bench, randomised parity cases, size scans.
set -u
REPO="${GRAFT_REPO_ROOT:-/root/repo}"
cd "$REPO"
O=gpurun_out/r3n; mkdir -p $O
export TMPDIR=/tmp
STEPS="${1:-smoke,test,bench,configs,roofline,stats,fuzz,scan}"
if [[ "$STEPS" == *smoke* ]]; then timeout 600 python -c "import __graft_entry__ as g; g.smoke()" > $O/smoke.log 2>&1; echo "smoke rc=$?"; tail -2 $O/smoke.log; fi
if [[ "$STEPS" == *test* ]]; then timeout 2400 python -m pytest tests -m gpu -q -p no:cacheprovider > $O/pytest_gpu.log 2>&1; echo "pytest rc=$?"; tail -4 $O/pytest_gpu.log; fi
if [[ "$STEPS" == *bench* ]]; then timeout 900 python bench.py --steps 20 --warmup 5 > $O/bench.json 2> $O/bench.err; echo "bench rc=$?"; head -c 400 $O/bench.json; echo; fi
if [[ "$STEPS" == *configs* ]]; then timeout 600 python tools/bench_configs.py 3 4 5 > $O/configs.jsonl 2> $O/configs.err; echo "configs rc=$?"; cut -c1-400 $O/configs.jsonl; timeout 300 python bench.py --workload config4 --steps 20 --warmup 5 --cg-iters 50 > $O/bench_config4.json 2>> $O/bench.err; timeout 300 python bench.py --workload slab --size 512 --steps 5 --warmup 2 > $O/bench_slab512.json 2>> $O/bench.err; echo "config4/slab rc=$?"; head -c 300 $O/bench_slab512.json; echo; fi
if [[ "$STEPS" == *roofline* ]]; then bash tools/kernel_roofline.sh $O/roofline; fi
if [[ "$STEPS" == *stats* ]]; then
  (cd /tmp && timeout 600 rocprofv3 --kernel-trace --stats --output-format csv -d "$REPO/$O/prof_bench" -o bench -- python "$REPO/bench.py" --steps 3 --warmup 1 --cpu-size 0 --profile-steps 0 --pmc 0 > "$REPO/$O/rocprof_bench.log" 2>&1); echo "stats rc=$?"
  find $O/prof_bench -name "*kernel_trace.csv" -size +2M -delete
fi
if [[ "$STEPS" == *fuzz* ]]; then timeout 1500 python tests/fuzz_parity.py --first 12000 --count 120 > $O/fuzz.log 2>&1; tail -1 $O/fuzz.log; grep -c "^ok" $O/fuzz.log; grep "^FAIL" $O/fuzz.log | head -5; fi
if [[ "$STEPS" == *scan* ]]; then
  timeout 600 python tools/size_scan.py --sizes 128,160,192,224,256,288,320,384,448,512 > $O/size_scan.jsonl 2> $O/scan.err
  timeout 600 python tools/size_scan.py --sizes 128,192,256,320,384 --dtype f64 > $O/size_scan_f64.jsonl 2>> $O/scan.err
  python - <<PY
import json
for f in ('$O/size_scan.jsonl','$O/size_scan_f64.jsonl'):
    for l in open(f):
        d=json.loads(l); t=d.get('tuned') or d['model']; print(d['size'], d['dtype'], 'it', t['us_iteration'], 'GB/s', t['moved_GBs_iteration'], 'mv', t['moved_GBs_matvec'])
PY
fi
