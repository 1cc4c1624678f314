#!/bin/bash
# Round-5 GPU session Q: does the sawtooth pay with the AUTOTUNED plans at 512^3? Six fresh contexts (= six first-call autotunes) per build, alternating
set -u
REPO="${GRAFT_REPO_ROOT:-/root/repo}"; cd "$REPO"
O=gpurun_out/${SESSION_TAG:-r5q}; mkdir -p $O; export TMPDIR=/tmp
for ROUND in 1 2 3 4 5 6; do for LIB in phiflow_amd/lib/libphihip_plain.so phiflow_amd/lib/libphihip.so; do
  timeout 120 python tools/size_scan.py --sizes 512,384 --iters 60 --lib $LIB 2>> $O/scan.err | sed "s|^{|{\"lib\": \"$(basename $LIB .so)\", |" >> $O/scan.jsonl
done; done
python - <<PY
import json
rows=[json.loads(l) for l in open('$O/scan.jsonl') if l.startswith('{')]
for r in rows:
    t=r['tuned']; print(r['lib'][9:].ljust(6) or 'saw', r['size'], round(t['us_iteration'],1), t['plan_mv'][:4], t['plan_x2'][:4], t['plan_ur'][:4], round(t['us_matvec'],1), round(t['us_update_x2'],1), round(t['us_update_r'],1))
PY
echo finished
