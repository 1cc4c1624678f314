#!/bin/bash
# same-box A/B of two builds of the library: tools/size_scan.py alternately with each (box-to-box spread is +-5 %, so only same-call
# comparisons count).  usage: tools/ab_lib.sh <other.so> [sizes] [tag]
REPO="${GRAFT_REPO_ROOT:-/root/repo}"; cd "$REPO"; mkdir -p gpurun_out; export TMPDIR=/tmp
OTHERS=${1:-phiflow_amd/lib/libphihip_prev.so}; SIZES=${2:-256,384,512}; TAG=${3:-ab}   # OTHERS: comma-separated list
OUT=gpurun_out/r02_${TAG}.jsonl; : > $OUT
for ROUND in 1 2; do
  for OTHER in ${OTHERS//,/ }; do
    timeout 300 python tools/size_scan.py --sizes $SIZES ${DTYPE:+--dtype $DTYPE} --lib $OTHER >> $OUT 2>> gpurun_out/r02_${TAG}.err
  done
  timeout 300 python tools/size_scan.py --sizes $SIZES ${DTYPE:+--dtype $DTYPE} >> $OUT 2>> gpurun_out/r02_${TAG}.err
done
python - $OUT <<'PY'
import json,sys
for l in open(sys.argv[1]):
    d=json.loads(l); print(d['lib'][:18].ljust(18), d['size'], {k:(v['us_matvec'],v['us_update_x2'],v['us_update_r'],v['us_iteration'],v['moved_GBs_iteration']) for k,v in d.items() if isinstance(v,dict)})
PY
