#!/bin/bash
# Round-5 GPU session I: smoke256 with the reach of the LDS-staged advection passes FIXED to 1 / 2 and adaptive (what should the policy's thresholds be?)
set -u
REPO="${GRAFT_REPO_ROOT:-/root/repo}"; cd "$REPO"
O=gpurun_out/${SESSION_TAG:-r5i}; mkdir -p $O; export TMPDIR=/tmp
for ROUND in 1 2; do for H in 1 2 -1 0; do for W in 30 90 150; do
    timeout 300 python bench.py --workload smoke256 --steps 40 --warmup $W --advect-halo $H > $O/tmp.json 2>> $O/err.log
    python - <<PY >> $O/smoke256_reach.jsonl
import json
d=json.load(open('$O/tmp.json'))
print(json.dumps({"halo": $H, "warmup": $W, "steps": 40, "ms_per_step": round(d["ms_per_step"],4), "op_ms_profiled_step": d.get("op_ms_profiled_step"), "fallback_last": d.get("advect_fallback_last_call")}))
PY
done; done; done
python - <<PY
import json
for l in open('$O/smoke256_reach.jsonl'):
    d=json.loads(l); o=d['op_ms_profiled_step']
    print('halo', d['halo'], 'warmup', d['warmup'], 'ms/step', d['ms_per_step'], 'mc_smoke', o['mac_cormack_smoke'], 'sl_v', o['semi_lagrangian_v'], 'fallback', d['fallback_last'])
PY
