#!/bin/bash
# Round-4 GPU session U: polling pause of the resident solver's waits (s_sleep 0 / 1 / 4 / 16 / 64 between polls) -- 256 workgroups polling through
# the fabric may slow each other down (8 x 512^2 runs 11.5 us per iteration against 7.5 for one entry)
set -u
REPO="${GRAFT_REPO_ROOT:-/root/repo}"
cd "$REPO"
O=gpurun_out/r4u; mkdir -p $O
export TMPDIR=/tmp PHIHIP_SWEEP_SHORT=1 PHIHIP_SWEEP_LIB=phiflow_amd/lib/libphihip_exp.so
: > $O/sweep_pause.jsonl
for ROUND in 1 2; do
  for PAUSE in 1 0 4 16 64; do
    PHIHIP_RES_PAUSE=$PAUSE timeout 300 python tools/sweep_resident.py 400 | sed "s/^{/{\"pause\": $PAUSE, /" >> $O/sweep_pause.jsonl 2>> $O/sweep_pause.err
  done
done
python - <<PY
import json
for l in open('$O/sweep_pause.jsonl'):
    d=json.loads(l)
    print('pause', d['pause'], d['res'], d['batch'], 'launches', d['launches']['us_per_iteration'], 'resident', d['resident']['us_per_iteration'], 'tol ms', d['resident']['tolerance_solve']['ms'])
PY
