#!/bin/bash
# Round-5 GPU session J: the adaptive reach after its re-calibration (12 % / 30 %, reach-tagged counts, bounded lag) against the fixed reaches
set -u
REPO="${GRAFT_REPO_ROOT:-/root/repo}"; cd "$REPO"
O=gpurun_out/${SESSION_TAG:-r5j}; mkdir -p $O; export TMPDIR=/tmp
python -c "from phiflow_amd import _capi as C; l=C.load_default_library(); print('build', l.build_id(), 'tree', l.built_from_tree())"
for CFL in 0.5 1.05 1.5 2.2; do for HALO in -1 1 2 0; do
  timeout 300 python tools/time_frow.py --size 256 --dtype f32 --bc closed --cfl $CFL --halo $HALO --only advect_self,mac_cormack_self,advect_centered,mac_cormack_centered --reps 130 >> $O/reach.jsonl 2>> $O/err.log
done; done
python - <<PY
import json
for l in open('$O/reach.jsonl'):
    d=json.loads(l); k=d['kernels']
    fb=d.get('advect_fallback'); print(d['bc'][:4], 'cfl', d['cfl'], 'halo', d['halo'], ' '.join(f"{n[:12]}={k[n].get('ms')}" for n in k), 'fallback', fb)
PY
for H in -1 1 2; do for W in 30 90 150; do
    timeout 300 python bench.py --workload smoke256 --steps 40 --warmup $W --advect-halo $H > $O/tmp.json 2>> $O/err.log
    python - <<PY >> $O/smoke256_reach.jsonl
import json
d=json.load(open('$O/tmp.json'))
print(json.dumps({"halo": $H, "warmup": $W, "steps": 40, "ms_per_step": round(d["ms_per_step"],4), "op_ms_profiled_step": d.get("op_ms_profiled_step"), "fallback_last": d.get("advect_fallback_last_call")}))
PY
done; done
python - <<PY
import json
for l in open('$O/smoke256_reach.jsonl'):
    d=json.loads(l); o=d['op_ms_profiled_step']
    print('halo', d['halo'], 'warmup', d['warmup'], 'ms/step', d['ms_per_step'], 'mc_smoke', o['mac_cormack_smoke'], 'sl_v', o['semi_lagrangian_v'], 'fallback', d['fallback_last'])
PY
