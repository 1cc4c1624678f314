#!/bin/bash
# mid-size dip, part 2: arbitrary chunk lengths (fill the resident slots exactly) and the distance-2 prefetch build (libphihip_pf2.so)
REPO="${GRAFT_REPO_ROOT:-/root/repo}"; cd "$REPO"; mkdir -p gpurun_out; export TMPDIR=/tmp
OUT=gpurun_out/r02_midsize2.jsonl; : > $OUT
CF1="1,16,24;1,16,32;1,16,39;1,16,43;1,16,48;1,16,55;1,16,64;1,16,77;1,16,96;1,16,128;2,32,32;2,32,39;2,32,48;2,32,64;2,32,96;2,16,48;2,16,64;4,64,48;4,64,64;4,64,96;4,32,64;1,64,48"
for LIB in "" phiflow_amd/lib/libphihip_pf2.so; do
  for FAM in 1 3; do
    PHIHIP_AUTOTUNE=0 timeout 300 python tools/sweep_cg.py --size 384 --family $FAM --iters 20 --configs "$CF1" ${LIB:+--lib $LIB} >> $OUT 2>> gpurun_out/r02_midsize2.err
  done
done
for LIB in "" phiflow_amd/lib/libphihip_pf2.so; do
  timeout 300 python tools/size_scan.py --sizes 256,320,384,448,512 ${LIB:+--lib $LIB} >> gpurun_out/r02_midsize2_scan.jsonl 2>> gpurun_out/r02_midsize2.err
done
python - <<'PY'
import json
for l in open('gpurun_out/r02_midsize2.jsonl'):
    d=json.loads(l); print(d['lib'][:14], d['family'], d['rows'], d['tpr'], d['chunk'], 'mv', d['ms_matvec'], 'ur', d['ms_update_r'], 'x2', d['ms_update'])
for l in open('gpurun_out/r02_midsize2_scan.jsonl'):
    d=json.loads(l); print(d['lib'][:14], d['size'], {k:(v['us_matvec'],v['us_update_x2'],v['us_update_r'],v['us_iteration'],v['plan_mv'][:3]) for k,v in d.items() if isinstance(v,dict)})
PY
