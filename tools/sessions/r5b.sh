#!/bin/bash
# Round-5 GPU session B: issue-rate microbenchmark (VALU classes, packed fp32, LDS read forms, LDS-DMA from 4-byte-aligned sources), the obstacle kernels of
# r5 (box-limited one-launch apply_obstacles / rasterisation, byte-parallel cell flags): GPU parity + same-box A/B against the round-4 library
set -u
REPO="${GRAFT_REPO_ROOT:-/root/repo}"; cd "$REPO"
O=gpurun_out/${SESSION_TAG:-r5b}; mkdir -p $O; export TMPDIR=/tmp
python -c "from phiflow_amd import _capi as C; l=C.load_default_library(); print('build', l.build_id(), 'tree', l.built_from_tree())" > $O/build_id.txt 2>&1; cat $O/build_id.txt
timeout 120 tools/micro/issue_rates > $O/issue_rates.txt 2>&1; echo "micro rc=$?"; cat $O/issue_rates.txt
timeout 900 python -m pytest tests/test_gpu_parity.py tests/test_gpu_api.py -m gpu -q -p no:cacheprovider -k "obstacle or cellflags or flags or incompressible or logo or wake or moving or cavity or batched" > $O/pytest_obst.log 2>&1; echo "obst rc=$?"; tail -3 $O/pytest_obst.log
for ROUND in 1 2; do for LIB in phiflow_amd/lib/libphihip_r4.so ""; do
  timeout 300 python tools/time_frow.py --size 256 --dtype f32 --bc closed --only obstacle_accessible,build_cellflags,apply_obstacles ${LIB:+--lib $LIB} >> $O/time_obst.jsonl 2>> $O/time_obst.err
done; done
timeout 300 python tools/time_frow.py --size 384 --dtype f64 --bc closed --only obstacle_accessible,build_cellflags,apply_obstacles --lib phiflow_amd/lib/libphihip_r4.so >> $O/time_obst.jsonl 2>> $O/time_obst.err
timeout 300 python tools/time_frow.py --size 384 --dtype f64 --bc closed --only obstacle_accessible,build_cellflags,apply_obstacles >> $O/time_obst.jsonl 2>> $O/time_obst.err
python - <<PY
import json
for l in open('$O/time_obst.jsonl'):
    d=json.loads(l)
    print(d['lib'][:16].ljust(16), d['size'], d['dtype'], d['bc'], ' '.join(f"{k}={v.get('ms','ERR')}" for k,v in d['kernels'].items()))
PY
