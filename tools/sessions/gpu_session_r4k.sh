#!/bin/bash
# Round-4 GPU session K (final): every record of profiles/r04_* from ONE build on ONE box -- GPU suite, smoke, the driver-style bench line, rocprofv3
# kernel stats of the bench step, the per-kernel roofline table (pinned plans), smoke256 / config4 bench lines (launch forms and the opt-in resident
# solver), BASELINE configs 3-5, same-box A/B against the round-3 library, the resident-solver sweep, randomised parity cases, L2 counters at 384^3 / 512^3
set -u
REPO="${GRAFT_REPO_ROOT:-/root/repo}"
cd "$REPO"
O=gpurun_out/${SESSION_TAG:-r4k}; mkdir -p $O
export TMPDIR=/tmp
python -c "from phiflow_amd import _capi as C; l=C.load_default_library(); print('build', l.build_id(), 'tree', l.built_from_tree())" > $O/build_id.txt 2>&1; cat $O/build_id.txt
timeout 1200 python -m pytest tests -m gpu -q -p no:cacheprovider > $O/pytest_gpu.log 2>&1; echo "pytest rc=$?"; tail -2 $O/pytest_gpu.log
timeout 300 python -c "import __graft_entry__ as g; g.smoke()" > $O/smoke.log 2>&1; echo "smoke rc=$?"; tail -1 $O/smoke.log
timeout 900 python bench.py --steps 20 --warmup 5 > $O/bench_n1.json 2> $O/bench_n1.err; echo "bench rc=$?"; head -c 300 $O/bench_n1.json; echo
timeout 300 bash tools/prof_bench_stats.sh ${SESSION_TAG:-r4k}/prof_bench > $O/prof_bench_summary.txt 2>&1; echo "prof_bench rc=$?"; head -6 $O/prof_bench_summary.txt
timeout 900 bash tools/kernel_roofline.sh $O/roofline > $O/roofline.log 2>&1; tail -3 $O/roofline.log
timeout 300 python bench.py --workload smoke256 --steps 20 --warmup 30 > $O/bench_smoke256.json 2> $O/bench_smoke256.err; echo "smoke256 rc=$?"
timeout 300 python bench.py --workload config4 --steps 20 --warmup 5 > $O/bench_config4.json 2> $O/bench_config4.err; echo "config4 rc=$?"
timeout 300 python bench.py --workload config4 --steps 20 --warmup 5 --resident-cg 2 > $O/bench_config4_resident.json 2> $O/bench_config4_resident.err; echo "config4 resident rc=$?"
python - <<PY
import json
for f in ('bench_smoke256','bench_config4','bench_config4_resident'):
    try:
        d=json.load(open('$O/'+f+'.json')); print(f, 'ms/step', round(d['ms_per_step'],4), 'us/it', d.get('us_per_cg_iteration_rank0'))
    except Exception as e: print(f, 'ERR', e)
PY
timeout 600 python tools/bench_configs.py 3 4 5 > $O/configs_345.jsonl 2> $O/configs_345.err; echo "configs rc=$?"; cut -c1-500 $O/configs_345.jsonl
: > $O/time_frow.jsonl
for ROUND in 1 2; do
  for LIB in phiflow_amd/lib/libphihip_r3.so ""; do
    timeout 300 python tools/time_frow.py --size 256 --dtype f32 --bc periodic ${LIB:+--lib $LIB} >> $O/time_frow.jsonl 2>> $O/time_frow.err
    timeout 300 python tools/time_frow.py --size 256 --dtype f32 --bc closed ${LIB:+--lib $LIB} >> $O/time_frow.jsonl 2>> $O/time_frow.err
    timeout 300 python tools/time_frow.py --size 384 --dtype f64 --bc closed ${LIB:+--lib $LIB} >> $O/time_frow.jsonl 2>> $O/time_frow.err
  done
done
for LIB in phiflow_amd/lib/libphihip_r3.so ""; do
  timeout 300 python tools/time_frow.py --size 512 --dtype f32 --bc periodic --reps 10 ${LIB:+--lib $LIB} >> $O/time_frow.jsonl 2>> $O/time_frow.err
  timeout 300 python tools/time_frow.py --size 256 --dtype f32 --bc periodic --cfl 1.5 ${LIB:+--lib $LIB} >> $O/time_frow.jsonl 2>> $O/time_frow.err
done
python - <<PY
import json
for l in open('$O/time_frow.jsonl'):
    d=json.loads(l)
    print(d['lib'][:16].ljust(16), d['size'], d['dtype'], d['bc'], 'cfl', d.get('cfl'), ' '.join(f"{k}={v.get('ms','ERR')}" for k,v in d['kernels'].items()), d.get('advect_fallback'))
PY
timeout 600 python tools/sweep_resident.py 400 > $O/sweep_resident.jsonl 2> $O/sweep_resident.err; echo "sweep rc=$?"
python - <<PY
import json
for l in open('$O/sweep_resident.jsonl'):
    d=json.loads(l)
    print(d['res'], d['batch'], d['bc'], 'launches', d['launches']['us_per_iteration'], 'resident', d['resident']['us_per_iteration'], 'x', d['speedup_resident'])
PY
timeout 600 python tests/fuzz_parity.py --first 42000 --count 40 > $O/fuzz.log 2>&1; tail -1 $O/fuzz.log; grep "^FAIL" $O/fuzz.log | head -5
# L2 view of the mid-size dip: hit / miss / fabric read requests of the CG kernels at 384^3 against 512^3 (one PMC pass each, no other trace domains)
for N in 384 512; do
  (cd /tmp && timeout 300 rocprofv3 --kernel-trace --pmc GRBM_GUI_ACTIVE TCC_HIT_sum TCC_MISS_sum TCC_EA0_RDREQ_sum --output-format csv -d "$REPO/$O/tcc_$N" -o pmc -- python "$REPO/tools/pmc_workload.py" $N > "$REPO/$O/tcc_$N.log" 2>&1); echo "tcc $N rc=$?"
  python tools/sq_summary.py $O/tcc_$N > $O/tcc_$N.json 2>/dev/null
done
(cd /tmp && timeout 120 rocprofv3 -L > "$REPO/$O/rocprofv3_counters.txt" 2>&1); grep -c "TCC_" $O/rocprofv3_counters.txt
find $O -name "*kernel_trace.csv" -size +2M -delete 2>/dev/null
find $O -name "*counter_collection.csv" -size +2M -delete 2>/dev/null
