#!/bin/bash
# Round-4 GPU session D: whole GPU suite on the current build, the smoke256 bench line, recognition cost on the device, the row-pitch
# experiment (boxes around the 288^3 ... 448^3 dip), and a first run of the pinned-plan profile tooling (one group)
set -u
REPO="${GRAFT_REPO_ROOT:-/root/repo}"
cd "$REPO"
O=gpurun_out/r4d; mkdir -p $O
export TMPDIR=/tmp
timeout 1200 python -m pytest tests -m gpu -q -p no:cacheprovider > $O/pytest_gpu.log 2>&1; echo "pytest rc=$?"; tail -4 $O/pytest_gpu.log
timeout 300 python bench.py --workload smoke256 --steps 20 --warmup 30 > $O/bench_smoke256.json 2> $O/bench_smoke256.err; echo "smoke256 rc=$?"; head -c 1500 $O/bench_smoke256.json; echo
timeout 300 python tools/time_recognition.py --sizes 64,128,256 > $O/time_recognition.jsonl 2> $O/time_recognition.err; echo "recognition rc=$?"; cat $O/time_recognition.jsonl
timeout 500 python tools/size_scan.py --iters 20 --sizes 288,288x288x320,288x320x288,320x288x288,320,384,384x384x416,384x416x384,416x384x384,448,448x448x480,448x448x512,512 > $O/scan_pitch.jsonl 2> $O/scan_pitch.err; echo "pitch rc=$?"
python - <<PY
import json
for l in open('$O/scan_pitch.jsonl'):
    d=json.loads(l); t=d.get('tuned') or d['model']; print(str(d['size']).ljust(14), 'it us', t['us_iteration'], 'GB/s', t['moved_GBs_iteration'], 'mv', t['moved_GBs_matvec'], t['plan_mv'], 'x2', t['moved_GBs_update_x2'], t['plan_x2'])
PY
timeout 600 bash tools/kernel_roofline.sh $O/roofline f32_256 > $O/roofline.log 2>&1; tail -5 $O/roofline.log
python - <<PY
import json
d=json.load(open('$O/roofline/kernel_roofline.json'))
for g in d['groups']:
    print(g['group'], g['build_id'], g.get('source_matches_tree'), g.get('cg_iteration_check'))
    for k in g['kernels']:
        if k.get('launches'): print(f"  {k['label'][:74]:74s} n={k['launches']:4d} {k['avg_us']:8.1f}us {k['frac_of_8TBs']:.3f} pmc/mv={k.get('pmc_over_moved')}")
PY
