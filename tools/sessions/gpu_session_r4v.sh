#!/bin/bash
# Round-4 GPU session V: resident solver with the boundary rows as plain 16-byte write-through vectors + one "ready" granule per (workgroup, side)
# (experimental build) against the tagged 8-byte granules that ship -- same box, alternating
set -u
REPO="${GRAFT_REPO_ROOT:-/root/repo}"
cd "$REPO"
O=gpurun_out/${SESSION_TAG:-r4v}; mkdir -p $O
export TMPDIR=/tmp PHIHIP_SWEEP_SHORT=1
: > $O/sweep_rows16.jsonl
for ROUND in 1 2; do
  for LIB in phiflow_amd/lib/libphihip_exp.so ""; do
    PHIHIP_SWEEP_LIB=$LIB timeout 300 python tools/sweep_resident.py 400 | sed "s#^{#{\"lib\": \"${LIB:-default}\", #" >> $O/sweep_rows16.jsonl 2>> $O/sweep_rows16.err
  done
done
python - <<PY
import json
for l in open('$O/sweep_rows16.jsonl'):
    d=json.loads(l)
    print(d['lib'][-16:].ljust(16), d['res'], d['batch'], 'launches', d['launches']['us_per_iteration'], 'resident', d['resident']['us_per_iteration'], 'tol ms', d['resident']['tolerance_solve']['ms'], d['resident']['tolerance_solve']['iterations'][:2], 'relL2 %.1e' % d['rel_l2_resident_vs_launches'])
PY
tail -3 $O/sweep_rows16.err
