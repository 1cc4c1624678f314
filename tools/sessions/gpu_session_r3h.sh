#!/bin/bash
# Round-3 GPU session H: (1) which half of the wall-split commit slowed the closed-box advection (A = before, B = HEAD, C = LDS patch without the
# split), (2) upper bounds for a tile without halo columns / without any halo (experimental builds, WRONG results by design: timing only),
# (3) price of the flag path at 384^3 fp64.
set -u
REPO="${GRAFT_REPO_ROOT:-/root/repo}"
cd "$REPO"
O=gpurun_out/r3h; mkdir -p $O
export TMPDIR=/tmp
E=$REPO/phiflow_amd/lib/exp
for REP in 1 2; do for L in $E/libphihip_advA.so "" $E/libphihip_advC.so; do
  LA=""; [ -n "$L" ] && LA="--lib $L"
  timeout 300 python tools/time_advect.py --size 256 --field tg --bc 1 $LA >> $O/time_advect.jsonl 2>> $O/adv.err
  timeout 300 python tools/time_advect.py --size 384 --dtype f64 --field tg --bc 1 $LA >> $O/time_advect.jsonl 2>> $O/adv.err
  timeout 300 python tools/time_advect.py --size 512 --field tg --bc 1 $LA >> $O/time_advect.jsonl 2>> $O/adv.err
done; done
python - <<PY
import json
for l in open('$O/time_advect.jsonl'):
    d=json.loads(l); print(d['lib'][:20], d['size'], d['dtype'], 'bc',d['bc'], d['ms_semi_lagrangian_staggered'], {k[-5:]:v for k,v in d.items() if k.startswith('ms_semi_lagrangian_staggered_halo')})
PY
for L in "" $E/libphihip_nohs.so $E/libphihip_nohalo.so; do
  LA=""; [ -n "$L" ] && LA="--lib $L"
  timeout 300 python tools/size_scan.py --sizes 288,384,448,512 $LA >> $O/scan.jsonl 2>> $O/scan.err
  timeout 300 python tools/size_scan.py --sizes 384 --dtype f64 --bc 1 $LA >> $O/scan.jsonl 2>> $O/scan.err
  timeout 300 python tools/size_scan.py --sizes 384 --dtype f64 --bc 1 --flags 1 $LA >> $O/scan.jsonl 2>> $O/scan.err
done
python - <<PY
import json
for l in open('$O/scan.jsonl'):
    d=json.loads(l)
    for lab in ('model','tuned'):
        if lab in d:
            t=d[lab]; print(d['lib'][:20], d['size'], d['dtype'], 'bc',d.get('bc'),'fl',d.get('flags'), lab, 'mv',t['us_matvec'],t['plan_mv'][:3],'x2',t['us_update_x2'],t['plan_x2'][:3],'ur',t['us_update_r'],t['plan_ur'][:3],'it GB/s',t['moved_GBs_iteration'])
PY
