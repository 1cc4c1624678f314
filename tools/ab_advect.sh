#!/bin/bash
# same-box A/B of the tiled advection across builds: tools/ab_advect.sh lib1.so,lib2.so   (the default build is always included)
REPO="${GRAFT_REPO_ROOT:-/root/repo}"; cd "$REPO"; mkdir -p gpurun_out; export TMPDIR=/tmp
OUT=gpurun_out/r02_ab_advect.jsonl; : > $OUT
for ROUND in 1 2; do
 for LIB in ${1//,/ } ""; do
  for ARGS in ${ADV_CASES:-"--size=256,--field=tg" "--size=256,--field=tg,--bc=1" "--size=512,--field=tg" "--size=384,--field=tg,--dtype=f64,--bc=1" "--size=256,--field=tg,--dtype=f64"}; do
    ARGS=${ARGS//,/ }
    timeout 200 python tools/time_advect.py $ARGS ${LIB:+--lib $LIB} >> $OUT 2>> gpurun_out/r02_ab_advect.err
  done
 done
done
python - <<'PY'
import json
for l in open('gpurun_out/r02_ab_advect.jsonl'):
    d=json.loads(l); print(d['lib'][:20].ljust(20), d['size'], d['dtype'], 'bc', d['bc'], 'halo1', d.get('ms_semi_lagrangian_staggered_halo1'), 'default', d.get('ms_semi_lagrangian_staggered'), 'chunks', {k[9:]:v for k,v in d.items() if k.startswith('ms_halo1_chunk')})
PY
