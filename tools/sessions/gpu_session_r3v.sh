#!/bin/bash
# Round-3 GPU session V: the V = 2 instantiation (ragged sizes), then the whole suite, the bench line and randomised cases on the final build
set -u
REPO="${GRAFT_REPO_ROOT:-/root/repo}"
cd "$REPO"
O=gpurun_out/r3v; mkdir -p $O
export TMPDIR=/tmp
timeout 300 python tools/size_scan.py --sizes 126,190,250,254,255,256 > $O/scan_ragged.jsonl 2> $O/scan.err
python - <<PY
import json
for l in open('$O/scan_ragged.jsonl'):
    d=json.loads(l); t=d.get('tuned') or d['model']; print(d['size'], 'it us', t['us_iteration'], 'GB/s', t['moved_GBs_iteration'], t['plan_mv'])
PY
timeout 600 python -c "import __graft_entry__ as g; g.smoke()" > $O/smoke.log 2>&1; echo "smoke rc=$?"; tail -1 $O/smoke.log
timeout 1500 python -m pytest tests -m gpu -q -p no:cacheprovider > $O/pytest_gpu.log 2>&1; echo "pytest rc=$?"; tail -3 $O/pytest_gpu.log
timeout 900 python bench.py --steps 20 --warmup 5 > $O/bench.json 2> $O/bench.err; echo "bench rc=$?"; head -c 300 $O/bench.json; echo
timeout 600 python tests/fuzz_parity.py --first 40000 --count 80 > $O/fuzz.log 2>&1; tail -1 $O/fuzz.log; grep -c "^ok" $O/fuzz.log; grep "^FAIL" $O/fuzz.log | head -5
