#!/bin/bash
# Round-4 GPU session L: rows that are not whole vectors on the UNAL kernels -- parity subset on the GPU, CG rate at 255^3 / 253^3 / 191^3 against
# 256^3 / 192^3 (and the same with the round-3 library: scalar kernels), odd fp64 rows
set -u
REPO="${GRAFT_REPO_ROOT:-/root/repo}"
cd "$REPO"
O=gpurun_out/r4l; mkdir -p $O
export TMPDIR=/tmp
python -c "from phiflow_amd import _capi as C; l=C.load_default_library(); print('build', l.build_id(), 'tree', l.built_from_tree())" > $O/build_id.txt 2>&1; cat $O/build_id.txt
timeout 900 python -m pytest tests/test_gpu_parity.py -m gpu -q -p no:cacheprovider -k "stencils or cg_matches or make_incompressible or obstacles or single_reduction or adaptive or implicit" -x > $O/pytest_subset.log 2>&1; echo "pytest rc=$?"; tail -3 $O/pytest_subset.log
: > $O/size_scan_odd.jsonl
for LIB in phiflow_amd/lib/libphihip_r3.so ""; do
  timeout 300 python tools/size_scan.py --sizes 256,255,253,254,192,191,320,319 ${LIB:+--lib $LIB} >> $O/size_scan_odd.jsonl 2>> $O/size_scan_odd.err
  timeout 300 python tools/size_scan.py --sizes 256,255,192,191 --dtype f64 ${LIB:+--lib $LIB} >> $O/size_scan_odd.jsonl 2>> $O/size_scan_odd.err
  timeout 300 python tools/size_scan.py --sizes 256,255 --bc 1 ${LIB:+--lib $LIB} >> $O/size_scan_odd.jsonl 2>> $O/size_scan_odd.err
done
python - <<PY
import json
for l in open('$O/size_scan_odd.jsonl'):
    d=json.loads(l)
    t=d.get('tuned') or d.get('model')
    print(d['lib'][:16].ljust(16), d['size'], d['dtype'], 'bc', d['bc'], 'us/it', t['us_iteration'], 'mv', t['us_matvec'], 'ur', t['us_update_r'], 'GB/s', t['moved_GBs_iteration'])
PY
