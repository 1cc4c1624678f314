#!/bin/bash
# Round-4 GPU session O: the ROW tile (whole rows of 65 ... 128 vectors, no halo columns) as an autotune candidate at 288^3 ... 480^3 -- same-box A/B
# against the round-3 library (two alternating rounds), forced row tile, parity subset
set -u
REPO="${GRAFT_REPO_ROOT:-/root/repo}"
cd "$REPO"
O=gpurun_out/r4o; mkdir -p $O
export TMPDIR=/tmp
python -c "from phiflow_amd import _capi as C; l=C.load_default_library(); print('build', l.build_id(), 'tree', l.built_from_tree())" > $O/build_id.txt 2>&1; cat $O/build_id.txt
timeout 900 python -m pytest tests/test_gpu_parity.py -m gpu -q -p no:cacheprovider -k "stencils or cg_matches or make_incompressible or obstacles or single_reduction or adaptive or implicit or tile" -x > $O/pytest_subset.log 2>&1; echo "pytest rc=$?"; tail -3 $O/pytest_subset.log
: > $O/size_scan_row_tile.jsonl
for ROUND in 1 2; do
  for LIB in phiflow_amd/lib/libphihip_r3.so ""; do
    timeout 600 python tools/size_scan.py --sizes 288,320,352,384,416,448,480,512 ${LIB:+--lib $LIB} >> $O/size_scan_row_tile.jsonl 2>> $O/size_scan_row_tile.err
  done
done
timeout 300 python tools/size_scan.py --sizes 192,256 --dtype f64 --lib phiflow_amd/lib/libphihip_r3.so >> $O/size_scan_row_tile.jsonl 2>> $O/size_scan_row_tile.err
timeout 300 python tools/size_scan.py --sizes 192,256 --dtype f64 >> $O/size_scan_row_tile.jsonl 2>> $O/size_scan_row_tile.err
python - <<PY
import json
for l in open('$O/size_scan_row_tile.jsonl'):
    d=json.loads(l)
    t=d.get('tuned') or d.get('model')
    print(d['lib'][:16].ljust(16), d['size'], d['dtype'], 'us/it', t['us_iteration'], 'mv', t['us_matvec'], 'ur', t['us_update_r'], 'x2', t.get('us_update_x2'), 'GB/s', t['moved_GBs_iteration'], 'plans mv', t.get('plan_mv'), 'ur', t.get('plan_ur'), 'x2', t.get('plan_x2'))
PY
