#!/bin/bash
# ONE parameterised GPU session script (r6; replaces the 62 one-off scripts of tools/sessions/ -- VERDICT r5 item 8):
#   gpurun --timeout 1800 -- 'bash tools/gpu_session.sh TAG stage [stage ...]'
# Every stage writes into gpurun_out/TAG/ and prints a short summary; a stage that fails does not stop the ones behind it. A stage may carry ONE argument after a colon
# (pytest:tests/test_jit.py, time_frow:384/f64/closed, fuzz:120). Summaries to keep are copied to profiles/ by tools/collect_profiles.sh TAG.
set -u
REPO="${GRAFT_REPO_ROOT:-/root/repo}"; cd "$REPO"
TAG="${1:?usage: gpu_session.sh TAG stage...}"; shift
O=gpurun_out/$TAG; mkdir -p $O; export TMPDIR=/tmp
LIBS="${AB_LIBS:-}"          # extra libraries for the A/B stages (space separated; HEAD always runs)

py() { python "$@"; }
jl() { python - "$@" <<'PY'
import json, sys
for l in open(sys.argv[1]):
    try: d = json.loads(l)
    except Exception: continue
    print(str(d.get('lib', 'HEAD'))[-18:].ljust(18), d.get('size'), d.get('dtype'), d.get('bc'), 'cfl', d.get('cfl'),
          ' '.join(f"{k}={v.get('ms', 'ERR')}" for k, v in d.get('kernels', {}).items()), d.get('advect_fallback'))
PY
}

stage_build_id() {
  py -c "from phiflow_amd import _capi as C; l=C.load_default_library(); print('build', l.build_id(), 'tree', l.built_from_tree())" > $O/build_id.txt 2>&1; cat $O/build_id.txt
}
stage_pytest() {            # pytest[:path or -k expression]   (default: the whole GPU suite)
  local what="${1:-tests}"; local name=$(echo "$what" | tr -c 'A-Za-z0-9' '_')
  eval "timeout 2400 python -m pytest $what -m gpu -q -p no:cacheprovider" > $O/pytest_$name.log 2>&1; echo "pytest [$what] rc=$?"; tail -3 $O/pytest_$name.log
}
stage_pytestx() {           # like pytest but stops at the first failure and prints it
  local what="${1:-tests}"; local name=$(echo "$what" | tr -c 'A-Za-z0-9' '_')
  eval "timeout 2400 python -m pytest $what -m gpu -q -x -p no:cacheprovider" > $O/pytest_$name.log 2>&1; echo "pytest -x [$what] rc=$?"; tail -30 $O/pytest_$name.log
}
stage_smoke() {
  timeout 300 python -c "import __graft_entry__ as g; g.smoke()" > $O/smoke.log 2>&1; echo "smoke rc=$?"; tail -1 $O/smoke.log
}
stage_bench() {             # the driver's own line
  timeout 900 python bench.py --steps 20 --warmup 5 > $O/bench_n1.json 2> $O/bench_n1.err; echo "bench rc=$?"; head -c 400 $O/bench_n1.json; echo
  py - <<PY
import json
try:
    d = json.loads(open('$O/bench_n1.json').read().strip().splitlines()[-1])
    print('ms_per_step', d['ms_per_step'], 'value', d['value'], 'roofline', {k: d['roofline'].get(k) for k in ('achieved', 'frac', 'traffic')}, 'config3', d.get('config3', {}).get('ms_per_iteration'))
    print('phi_level', json.dumps(d.get('phi_level'))[:600])
except Exception as e:
    print('bench line unreadable', e)
PY
}
stage_bench_quick() {       # the 256^3 line without the side measurements
  timeout 600 python bench.py --steps 20 --warmup 5 --pmc 0 --cpu-size 0 --phi-level 0 > $O/bench_quick.json 2> $O/bench_quick.err; echo "bench_quick rc=$?"
  py -c "import json;d=json.loads(open('$O/bench_quick.json').read().strip().splitlines()[-1]);print('ms_per_step', d['ms_per_step'], 'config3', d.get('config3',{}).get('ms_per_iteration'))"
}
stage_jit_debug() {         # the replay-after-_foreach_ record
  timeout 900 python tools/micro/jit_foreach_debug.py > $O/jit_foreach_debug.txt 2>&1; echo "jit_foreach_debug rc=$?"; grep "first difference" $O/jit_foreach_debug.txt
}
stage_prof_bench() {
  timeout 600 bash tools/prof_bench_stats.sh $TAG/prof_bench > $O/prof_bench_summary.txt 2>&1; echo "prof_bench rc=$?"; head -8 $O/prof_bench_summary.txt
}
stage_roofline() {
  timeout 1500 bash tools/kernel_roofline.sh $O/roofline > $O/roofline.log 2>&1; echo "roofline rc=$?"; tail -3 $O/roofline.log
}
stage_configs() {           # BASELINE configs 3-5
  timeout 900 python tools/bench_configs.py ${1:-3 4 5} > $O/configs.jsonl 2> $O/configs.err; echo "configs rc=$?"; cut -c1-500 $O/configs.jsonl
}
stage_time_frow() {         # time_frow:SIZE/DTYPE/BC[/CFL]  -- HEAD and every library of AB_LIBS, two alternating rounds
  local spec="${1:-256/f32/periodic}"; IFS=/ read -r size dt bc cfl <<< "$spec"
  local out=$O/time_frow_$(echo "$spec" | tr '/' '_').jsonl; : > $out
  for ROUND in 1 2; do for LIB in $LIBS ""; do
    timeout 400 python tools/time_frow.py --size $size --dtype $dt --bc $bc ${cfl:+--cfl $cfl} ${LIB:+--lib $LIB} >> $out 2>> $O/time_frow.err
  done; done
  jl $out
}
stage_smoke256() {
  for W in ${1:-30}; do
    timeout 400 python bench.py --workload smoke256 --steps 20 --warmup $W --pmc 0 --cpu-size 0 --phi-level 0 > $O/smoke256_w$W.json 2>> $O/smoke256.err
    py -c "import json;d=json.loads(open('$O/smoke256_w$W.json').read().strip().splitlines()[-1]);print('smoke256 warmup $W', d['ms_per_step'], d.get('op_ms_profiled_step'), d.get('advect_fallback_last_call'))"
  done
}
stage_config4() {           # the launch forms (0), the library default (-1) and "whenever applicable" (2)
  for R in 0 -1 2; do
    timeout 400 python bench.py --workload config4 --steps 20 --warmup 5 --resident-cg $R > $O/config4_res$R.json 2>> $O/config4.err
    py -c "import json;d=json.loads(open('$O/config4_res$R.json').read().strip().splitlines()[-1]);print('config4 resident $R ms/step', d['ms_per_step'], 'us/it', d.get('us_per_cg_iteration_rank0'))"
  done
}
stage_resident() {          # launch forms vs resident solver: the short sweep, then B x 512^2 on one GPU (configs[3]'s sharding story); resident:coop1 = cooperative launch
  local env="" tag="plain"; [ "${1:-}" = "coop1" ] && { env="PHIHIP_RESIDENT_COOP=1"; tag="coop1"; }
  env $env PHIHIP_SWEEP_SHORT=1 timeout 600 python tools/sweep_resident.py 400 > $O/sweep_resident_$tag.jsonl 2>> $O/sweep_resident.err
  env $env PHIHIP_SWEEP_BATCHES=1 timeout 900 python tools/sweep_resident.py 400 > $O/sweep_resident_batches_$tag.jsonl 2>> $O/sweep_resident.err
  env $env PHIHIP_SWEEP_SHORT=1 PHIHIP_SWEEP_FLAGS=1 timeout 600 python tools/sweep_resident.py 400 > $O/sweep_resident_flags_$tag.jsonl 2>> $O/sweep_resident.err
  py - <<PY
import json
for f in ('$O/sweep_resident_$tag.jsonl', '$O/sweep_resident_batches_$tag.jsonl', '$O/sweep_resident_flags_$tag.jsonl'):
    for l in open(f):
        d = json.loads(l)
        print(d['res'], 'x', d['batch'], d['bc'], 'launches', d['launches']['us_per_iteration'], 'resident', d['resident']['us_per_iteration'], 'speedup', d['speedup_resident'],
              'tol ms', d['launches']['tolerance_solve']['ms'], d['resident']['tolerance_solve']['ms'], 'rel', '%.1e' % d['rel_l2_resident_vs_launches'])
PY
}
stage_fuzz() {
  timeout 2400 python tests/fuzz_parity.py --first ${FUZZ_SEED:-60000} --count ${1:-120} > $O/fuzz.txt 2>&1; echo "fuzz rc=$?"; tail -4 $O/fuzz.txt
}
stage_backward() {
  timeout 600 python tools/time_backward_step.py > $O/backward_step.jsonl 2>&1; echo "backward rc=$?"; tail -2 $O/backward_step.jsonl | cut -c1-600
}
stage_size_scan() {
  timeout 900 python tools/size_scan.py ${1:-} > $O/size_scan.jsonl 2> $O/size_scan.err; echo "size_scan rc=$?"; cut -c1-300 $O/size_scan.jsonl
}
stage_run() {               # run:'command' -- anything else, output to run_N.log
  local n=$(ls $O/run_*.log 2>/dev/null | wc -l); timeout 1500 bash -c "$1" > $O/run_$n.log 2>&1; echo "run [$1] rc=$?"; tail -25 $O/run_$n.log
}

for st in "$@"; do
  name="${st%%:*}"; arg=""; [ "$name" != "$st" ] && arg="${st#*:}"
  echo "=== $name $arg"
  if declare -f "stage_$name" > /dev/null; then "stage_$name" ${arg:+"$arg"}; else echo "unknown stage $name"; fi
done
echo finished
