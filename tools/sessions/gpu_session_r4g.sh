#!/bin/bash
# Round-4 GPU session G: every record of profiles/r04_* from ONE build on ONE box -- GPU suite, smoke, the driver-style bench line, rocprofv3
# kernel stats of the bench step, the per-kernel roofline table (pinned plans), smoke256 / config4 bench lines, BASELINE configs 3-5,
# same-box A/B against the round-3 library, randomised parity cases
set -u
REPO="${GRAFT_REPO_ROOT:-/root/repo}"
cd "$REPO"
O=gpurun_out/r4g; mkdir -p $O
export TMPDIR=/tmp
python -c "from phiflow_amd import _capi as C; l=C.load_default_library(); print('build', l.build_id(), 'tree', l.built_from_tree())" > $O/build_id.txt 2>&1; cat $O/build_id.txt
timeout 1200 python -m pytest tests -m gpu -q -p no:cacheprovider > $O/pytest_gpu.log 2>&1; echo "pytest rc=$?"; tail -2 $O/pytest_gpu.log
timeout 300 python -c "import __graft_entry__ as g; g.smoke()" > $O/smoke.log 2>&1; echo "smoke rc=$?"; tail -1 $O/smoke.log
timeout 900 python bench.py --steps 20 --warmup 5 > $O/bench_n1.json 2> $O/bench_n1.err; echo "bench rc=$?"; head -c 400 $O/bench_n1.json; echo
timeout 300 bash tools/prof_bench_stats.sh r4g/prof_bench > $O/prof_bench_summary.txt 2>&1; echo "prof_bench rc=$?"; head -8 $O/prof_bench_summary.txt
timeout 900 bash tools/kernel_roofline.sh $O/roofline > $O/roofline.log 2>&1; tail -3 $O/roofline.log
timeout 300 python bench.py --workload smoke256 --steps 20 --warmup 30 > $O/bench_smoke256.json 2> $O/bench_smoke256.err; echo "smoke256 rc=$?"
timeout 300 python bench.py --workload config4 --steps 20 --warmup 5 > $O/bench_config4.json 2> $O/bench_config4.err; echo "config4 rc=$?"
timeout 600 python tools/bench_configs.py 3 4 5 > $O/configs_345.jsonl 2> $O/configs_345.err; echo "configs rc=$?"; cat $O/configs_345.jsonl | cut -c1-400
: > $O/time_frow.jsonl
for ROUND in 1 2; do
  for LIB in phiflow_amd/lib/libphihip_r3.so ""; do
    timeout 300 python tools/time_frow.py --size 256 --dtype f32 --bc periodic ${LIB:+--lib $LIB} >> $O/time_frow.jsonl 2>> $O/time_frow.err
    timeout 300 python tools/time_frow.py --size 384 --dtype f64 --bc closed ${LIB:+--lib $LIB} >> $O/time_frow.jsonl 2>> $O/time_frow.err
  done
done
for LIB in phiflow_amd/lib/libphihip_r3.so ""; do
  timeout 300 python tools/time_frow.py --size 256 --dtype f32 --bc closed ${LIB:+--lib $LIB} >> $O/time_frow.jsonl 2>> $O/time_frow.err
  timeout 300 python tools/time_frow.py --size 512 --dtype f32 --bc periodic --reps 10 ${LIB:+--lib $LIB} >> $O/time_frow.jsonl 2>> $O/time_frow.err
  timeout 300 python tools/time_frow.py --size 256 --dtype f32 --bc periodic --cfl 1.5 ${LIB:+--lib $LIB} >> $O/time_frow.jsonl 2>> $O/time_frow.err
done
python - <<PY
import json
for l in open('$O/time_frow.jsonl'):
    d=json.loads(l)
    print(d['lib'][:16].ljust(16), d['size'], d['dtype'], d['bc'], 'cfl', d.get('cfl'), ' '.join(f"{k}={v.get('ms','ERR')}" for k,v in d['kernels'].items()), d.get('advect_fallback'))
PY
timeout 600 python tests/fuzz_parity.py --first 41000 --count 40 > $O/fuzz.log 2>&1; tail -1 $O/fuzz.log; grep "^FAIL" $O/fuzz.log | head -5
find $O -name "*kernel_trace.csv" -size +2M -delete 2>/dev/null
