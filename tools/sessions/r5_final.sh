#!/bin/bash
# Round-5 FINAL GPU session: every record of profiles/r05_* that describes the shipped build comes from ONE build on ONE box -- GPU suite, smoke,
# the driver-style bench line, rocprofv3 kernel stats of the bench step, the per-kernel roofline table (pinned plans), smoke256 / config4 lines (launch
# forms and the opt-in resident solver), BASELINE configs 3-5, same-box A/B against the round-4 library, randomised parity cases (resident arm included),
# the issue-rate microbenchmark, the differentiated step. The A/B partner is built by tools/build_r4_library.sh (without it the A/B rows are HEAD only).   SESSION_TAG=r5z bash tools/sessions/r5_final.sh ; then tools/collect_profiles.sh r5z
set -u
REPO="${GRAFT_REPO_ROOT:-/root/repo}"; cd "$REPO"
O=gpurun_out/${SESSION_TAG:-r5z}; mkdir -p $O; export TMPDIR=/tmp
R4=phiflow_amd/lib/libphihip_r4.so; [ -f $R4 ] || R4=""
python -c "from phiflow_amd import _capi as C; l=C.load_default_library(); print('build', l.build_id(), 'tree', l.built_from_tree())" > $O/build_id.txt 2>&1; cat $O/build_id.txt
timeout 1500 python -m pytest tests -m gpu -q -p no:cacheprovider > $O/pytest_gpu.log 2>&1; echo "pytest rc=$?"; tail -2 $O/pytest_gpu.log
timeout 300 python -c "import __graft_entry__ as g; g.smoke()" > $O/smoke.log 2>&1; echo "smoke rc=$?"; tail -1 $O/smoke.log
timeout 900 python bench.py --steps 20 --warmup 5 > $O/bench_n1.json 2> $O/bench_n1.err; echo "bench rc=$?"; head -c 300 $O/bench_n1.json; echo
timeout 300 bash tools/prof_bench_stats.sh ${SESSION_TAG:-r5z}/prof_bench > $O/prof_bench_summary.txt 2>&1; echo "prof_bench rc=$?"; head -6 $O/prof_bench_summary.txt
timeout 1200 bash tools/kernel_roofline.sh $O/roofline > $O/roofline.log 2>&1; tail -3 $O/roofline.log
timeout 300 python bench.py --workload smoke256 --steps 20 --warmup 30 > $O/bench_smoke256.json 2> $O/bench_smoke256.err; echo "smoke256 rc=$?"
timeout 300 python bench.py --workload config4 --steps 20 --warmup 5 > $O/bench_config4.json 2> $O/bench_config4.err; echo "config4 rc=$?"
timeout 300 python bench.py --workload config4 --steps 20 --warmup 5 --resident-cg 2 > $O/bench_config4_resident.json 2> $O/bench_config4_resident.err; echo "config4 resident rc=$?"
python - <<PY
import json
for f in ('bench_smoke256','bench_config4','bench_config4_resident'):
    try:
        d=json.load(open('$O/'+f+'.json')); print(f, 'ms/step', round(d['ms_per_step'],4), 'us/it', d.get('us_per_cg_iteration_rank0'), d.get('non_cg_share_of_profiled_step'))
    except Exception as e: print(f, 'ERR', e)
PY
timeout 600 python tools/bench_configs.py 3 4 5 > $O/configs_345.jsonl 2> $O/configs_345.err; echo "configs rc=$?"; cut -c1-400 $O/configs_345.jsonl
: > $O/time_frow.jsonl
for ROUND in 1 2; do
  for LIB in $R4 ""; do
    timeout 300 python tools/time_frow.py --size 256 --dtype f32 --bc periodic ${LIB:+--lib $LIB} >> $O/time_frow.jsonl 2>> $O/time_frow.err
    timeout 300 python tools/time_frow.py --size 256 --dtype f32 --bc closed ${LIB:+--lib $LIB} >> $O/time_frow.jsonl 2>> $O/time_frow.err
    timeout 300 python tools/time_frow.py --size 384 --dtype f64 --bc closed ${LIB:+--lib $LIB} >> $O/time_frow.jsonl 2>> $O/time_frow.err
  done
done
for LIB in $R4 ""; do
  timeout 300 python tools/time_frow.py --size 512 --dtype f32 --bc periodic --reps 10 ${LIB:+--lib $LIB} >> $O/time_frow.jsonl 2>> $O/time_frow.err
  timeout 300 python tools/time_frow.py --size 256 --dtype f32 --bc periodic --cfl 1.5 ${LIB:+--lib $LIB} >> $O/time_frow.jsonl 2>> $O/time_frow.err
done
python - <<PY
import json
for l in open('$O/time_frow.jsonl'):
    d=json.loads(l)
    print(d['lib'][:16].ljust(16), d['size'], d['dtype'], d['bc'], 'cfl', d.get('cfl'), ' '.join(f"{k}={v.get('ms','ERR')}" for k,v in d['kernels'].items()), d.get('advect_fallback'))
PY
timeout 120 tools/micro/issue_rates > $O/issue_rates.txt 2>&1; echo "micro rc=$?"
# smoke256 as a same-box A/B against the round-4 library at three stages of the plume (where the one-off switch of the adaptive reach falls decides a single line)
: > $O/smoke256_ab.jsonl
for ROUND in 1 2; do for LIB in $R4 ""; do for W in 30 90 150; do
    PHIHIP_LIBRARY=$LIB timeout 300 python bench.py --workload smoke256 --steps 40 --warmup $W > $O/tmp.json 2>> $O/smoke256_ab.err
    python - <<PY >> $O/smoke256_ab.jsonl
import json
d=json.load(open('$O/tmp.json'))
print(json.dumps({"lib": "$LIB" or "HEAD", "warmup": $W, "steps": 40, "ms_per_step": round(d["ms_per_step"],4), "op_ms_profiled_step": d.get("op_ms_profiled_step"), "non_cg_share": d.get("non_cg_share_of_profiled_step"), "fallback": d.get("advect_fallback_last_call"), "build_id": d.get("build_id")}))
PY
done; done; done
python - <<PY
import json
for l in open('$O/smoke256_ab.jsonl'):
    d=json.loads(l); o=d['op_ms_profiled_step']; print(d['lib'][-14:].ljust(14), 'warmup', d['warmup'], 'ms/step', d['ms_per_step'], 'mc_smoke', o['mac_cormack_smoke'], 'sl_v', o['semi_lagrangian_v'], d['fallback'])
PY
timeout 600 python tools/time_backward_step.py > $O/backward_step.jsonl 2> $O/backward_step.err; echo "bwd rc=$?"; cut -c1-300 $O/backward_step.jsonl
timeout 300 python tools/time_host_api.py --size 128 > $O/host_api.jsonl 2>> $O/host_api.err; timeout 300 python tools/time_host_api.py --size 512 --batch 8 >> $O/host_api.jsonl 2>> $O/host_api.err; cat $O/host_api.jsonl
find $O -name "*kernel_trace.csv" -size +2M -delete 2>/dev/null
find $O -name "*counter_collection.csv" -size +2M -delete 2>/dev/null
