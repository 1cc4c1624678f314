#!/bin/bash
# Round-5 GPU session O: sawtooth marching with ONE chunk length for all three kernel families (do the far ends of the chunks coincide?) -- pinned
# (rows, tpr, chunk) shared by MATVEC / UPDATE_R / UPDATE_X2, plain order against sawtooth, per size
set -u
REPO="${GRAFT_REPO_ROOT:-/root/repo}"; cd "$REPO"
O=gpurun_out/${SESSION_TAG:-r5o}; mkdir -p $O; export TMPDIR=/tmp
CFG="1,64,16;1,64,32;1,64,64;1,64,128;1,32,32;1,32,64;2,64,32;2,64,64;4,64,64;1,16,32;1,16,64"
for N in 256 320 384 512; do for LIB in phiflow_amd/lib/libphihip_plain.so phiflow_amd/lib/libphihip.so; do
  timeout 300 python tools/sweep_cg.py --size $N --iters 40 --configs "$CFG" --lib $LIB >> $O/sweep.jsonl 2>> $O/sweep.err
done; done
python - <<PY
import json
rows=[json.loads(l) for l in open('$O/sweep.jsonl') if l.startswith('{')]
by={}
for r in rows: by.setdefault((r['size'], tuple(r['plan_mv'][:3]) if r['plan_mv'] else None, r['plan_mv'][3] if r['plan_mv'] else 0),{})[r['lib']]=r['ms_iter_wall']
last=None
for k,v in by.items(): print(k, v)
PY
echo finished
