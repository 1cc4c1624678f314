#!/bin/bash
# Round-3 GPU session Q: gather-form advection adjoints (tests, timing under rocprofv3 next to the atomic version of the previous build)
set -u
REPO="${GRAFT_REPO_ROOT:-/root/repo}"
cd "$REPO"
O=gpurun_out/r3q; mkdir -p $O
export TMPDIR=/tmp
timeout 900 python -m pytest tests -m gpu -q -p no:cacheprovider -k "adjoint or gradient or tutorial" > $O/pytest_adj.log 2>&1; echo "pytest rc=$?"; tail -3 $O/pytest_adj.log
(cd /tmp && timeout 600 rocprofv3 --kernel-trace --stats --output-format csv -d "$REPO/$O/stats" -o k -- python "$REPO/tools/path_workload.py" --group f32_256 --reps 8 > "$REPO/$O/stats.log" 2>&1); echo "stats rc=$?"
python - <<PY
import csv,glob
from collections import defaultdict
per=defaultdict(list)
for f in glob.glob('$O/stats/**/*kernel_trace.csv', recursive=True):
    for r in csv.DictReader(open(f)):
        n=r['Kernel_Name']
        if 'bwd' in n: per[n.split('(')[0].replace('void phihip::','')].append((int(r['End_Timestamp'])-int(r['Start_Timestamp']))/1e3)
for n,t in sorted(per.items()):
    t.sort(); print(f"{n[:70]:70s} calls {len(t):3d} min {t[0]:8.1f} median {t[len(t)//2]:8.1f} max {t[-1]:8.1f}")
PY
timeout 600 python tests/fuzz_parity.py --first 13000 --count 40 > $O/fuzz.log 2>&1; tail -1 $O/fuzz.log; grep "^FAIL" $O/fuzz.log | head -5
