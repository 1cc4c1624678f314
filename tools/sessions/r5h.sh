#!/bin/bash
# Round-5 GPU session H: where is the cross-over between the narrow reach + fix-up list and the wide reach? (the adaptive policy switches at 2 % of the units)
set -u
REPO="${GRAFT_REPO_ROOT:-/root/repo}"; cd "$REPO"
O=gpurun_out/${SESSION_TAG:-r5h}; mkdir -p $O; export TMPDIR=/tmp
for BC in closed periodic; do for CFL in 0.9 1.05 1.15 1.3 1.5 1.8 2.2; do for HALO in 1 2 0; do
  timeout 300 python tools/time_frow.py --size 256 --dtype f32 --bc $BC --cfl $CFL --halo $HALO --only advect_self,mac_cormack_self,advect_centered,mac_cormack_centered --reps 20 >> $O/reach.jsonl 2>> $O/err.log
done; done; done
python - <<PY
import json
for l in open('$O/reach.jsonl'):
    d=json.loads(l); k=d['kernels']
    fb=d.get('advect_fallback'); print(d['bc'][:4], 'cfl', d['cfl'], 'halo', d['halo'], ' '.join(f"{n[:12]}={k[n].get('ms')}" for n in k), 'fallback', fb, round(fb[0]/max(1,fb[1]),3) if fb else None)
PY
