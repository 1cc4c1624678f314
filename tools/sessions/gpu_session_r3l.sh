#!/bin/bash
# Round-3 GPU session L: gather kernels with K samples per thread (tests + timing against the round-2 library)
set -u
REPO="${GRAFT_REPO_ROOT:-/root/repo}"
cd "$REPO"
O=gpurun_out/r3l; mkdir -p $O
export TMPDIR=/tmp
timeout 900 python -m pytest tests -m gpu -q -p no:cacheprovider -k "advect or mac_cormack or golden or adjoint or different_grids or reference_style" > $O/pytest.log 2>&1; echo "pytest rc=$?"; tail -3 $O/pytest.log
for REP in 1 2; do for L in ab/libphihip_base.so ""; do
  LA=""; [ -n "$L" ] && LA="--lib $REPO/$L"
  timeout 300 python tools/time_advect.py --size 256 --field tg $LA >> $O/time_advect.jsonl 2>> $O/adv.err
  timeout 300 python tools/time_advect.py --size 256 --field tg --bc 1 $LA >> $O/time_advect.jsonl 2>> $O/adv.err
  timeout 300 python tools/time_advect.py --size 256 --dtype f64 --field tg --bc 1 $LA >> $O/time_advect.jsonl 2>> $O/adv.err
  timeout 300 python tools/time_advect.py --size 512 --field tg $LA >> $O/time_advect.jsonl 2>> $O/adv.err
done; done
python - <<PY
import json
for l in open('$O/time_advect.jsonl'):
    d=json.loads(l); print(d['lib'][:12], d['size'], d['dtype'], 'bc',d['bc'], 'SL stag',d['ms_semi_lagrangian_staggered'], 'MC stag', d['ms_mac_cormack_staggered'], 'SL cen', d['ms_semi_lagrangian_centered'], 'MC cen', d['ms_mac_cormack_centered'])
PY
