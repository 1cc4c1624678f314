#!/bin/bash
# Round-5 GPU session S: deterministic adaptive reach (one named pass, read 4 passes later) -- jit tests, the CFL 1.5 timings and the smoke256 line with it
set -u
REPO="${GRAFT_REPO_ROOT:-/root/repo}"; cd "$REPO"
O=gpurun_out/${SESSION_TAG:-r5s}; mkdir -p $O; export TMPDIR=/tmp
python -c "from phiflow_amd import _capi as C; l=C.load_default_library(); print('build', l.build_id(), 'tree', l.built_from_tree())" > $O/build_id.txt 2>&1; cat $O/build_id.txt
timeout 600 python -m pytest tests/test_jit.py tests/test_gpu_graph.py -m gpu -q -p no:cacheprovider > $O/pytest_jit.log 2>&1; echo "jit rc=$?"; tail -4 $O/pytest_jit.log
timeout 900 python -m pytest tests/test_gpu_parity.py -m gpu -q -p no:cacheprovider -x -k "advect or advection or mac_cormack" > $O/pytest_adv.log 2>&1; echo "adv rc=$?"; tail -2 $O/pytest_adv.log
for CFL in 0.5 1.5; do
  timeout 300 python tools/time_frow.py --size 256 --dtype f32 --bc periodic --cfl $CFL --only advect_self,mac_cormack_self,advect_centered,mac_cormack_centered --reps 40 >> $O/time.jsonl 2>> $O/time.err
done
python - <<PY
import json
for l in open('$O/time.jsonl'):
    d=json.loads(l); print(d['size'], d['bc'], 'cfl', d.get('cfl'), ' '.join(f"{k}={v.get('ms','ERR')}" for k,v in d['kernels'].items()), d.get('advect_fallback'))
PY
for W in 30 90; do
  timeout 300 python bench.py --workload smoke256 --steps 20 --warmup $W --pmc 0 --cpu-size 0 --phi-level 0 > $O/smoke256_w$W.json 2>> $O/smoke.err
  python -c "import json;d=json.loads(open('$O/smoke256_w$W.json').read().strip().splitlines()[-1]);print('smoke256 warmup $W', d['ms_per_step'], d['op_ms_profiled_step']['mac_cormack_smoke'], d['op_ms_profiled_step']['semi_lagrangian_v'], d['advect_fallback_last_call'])"
done
echo finished
