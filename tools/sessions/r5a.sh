#!/bin/bash
# Round-5 GPU session A: the new pins only (row tiles, resident solver in config 4 + fuzz arm, two ranks on one GPU) + a timing baseline of the
# advection family (HEAD vs the round-4 library) for the kernel work that follows
set -u
REPO="${GRAFT_REPO_ROOT:-/root/repo}"; cd "$REPO"
O=gpurun_out/${SESSION_TAG:-r5a}; mkdir -p $O; export TMPDIR=/tmp
python -c "from phiflow_amd import _capi as C; l=C.load_default_library(); print('build', l.build_id(), 'tree', l.built_from_tree())" > $O/build_id.txt 2>&1; cat $O/build_id.txt
timeout 900 python -m pytest tests/test_gpu_parity.py -m gpu -q -p no:cacheprovider -k "row_tiles or tile_configurations or resident" > $O/pytest_rowt.log 2>&1; echo "rowt rc=$?"; tail -3 $O/pytest_rowt.log
timeout 600 python -m pytest tests/test_gpu_baseline_sizes.py -m gpu -q -p no:cacheprovider -k "config4" > $O/pytest_c4.log 2>&1; echo "c4 rc=$?"; tail -3 $O/pytest_c4.log
timeout 600 python -m pytest tests/test_slab_two_ranks.py -m gpu -q -p no:cacheprovider -x > $O/pytest_two_ranks.log 2>&1; echo "two_ranks rc=$?"; tail -5 $O/pytest_two_ranks.log
timeout 600 python tests/fuzz_parity.py --first 51000 --count 30 > $O/fuzz.log 2>&1; tail -1 $O/fuzz.log; grep "^FAIL" $O/fuzz.log | head -5
for LIB in phiflow_amd/lib/libphihip_r4.so ""; do
  timeout 300 python tools/time_frow.py --size 256 --dtype f32 --bc periodic ${LIB:+--lib $LIB} >> $O/time_frow.jsonl 2>> $O/time_frow.err
  timeout 300 python tools/time_frow.py --size 256 --dtype f32 --bc closed ${LIB:+--lib $LIB} >> $O/time_frow.jsonl 2>> $O/time_frow.err
done
python - <<PY
import json
for l in open('$O/time_frow.jsonl'):
    d=json.loads(l)
    print(d['lib'][:16].ljust(16), d['size'], d['dtype'], d['bc'], ' '.join(f"{k}={v.get('ms','ERR')}" for k,v in d['kernels'].items()), d.get('advect_fallback'))
PY
# SQ / TCC counters of the self-advection at 256^3 (which unit bounds it: VALU issue, LDS, VMEM address, waits) -- the ISA audit's other half
timeout 400 bash tools/prof_advect.sh "--size 256 --field tg" > $O/prof_advect.log 2>&1; cp gpurun_out/prof_adv_pmc.json $O/ 2>/dev/null; tail -3 $O/prof_advect.log | cut -c1-300
find gpurun_out/prof_adv_pmc gpurun_out/prof_adv_stats -name "*.csv" -size +1M -delete 2>/dev/null
