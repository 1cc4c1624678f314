#!/bin/bash
# Round-3 GPU session I: adjoint kernels with LDS windows (tests + roofline group f32_256), advection after the revert of the wall split.
set -u
REPO="${GRAFT_REPO_ROOT:-/root/repo}"
cd "$REPO"
O=gpurun_out/r3i; mkdir -p $O
export TMPDIR=/tmp
timeout 900 python -m pytest tests -m gpu -q -p no:cacheprovider -k "adjoint or gradient or tutorial" > $O/pytest_adj.log 2>&1; echo "pytest rc=$?"; tail -3 $O/pytest_adj.log
bash tools/kernel_roofline.sh $O/roofline f32_256; python - <<PY
import json
d=json.load(open('$O/roofline/kernel_roofline.json'))
for g in d['groups']:
    for k in g['kernels']:
        if k.get('launches',0): print(f"{g['group']} {k['label'][:58]:58s} {k['avg_us']:8.1f} us frac {k['frac_of_8TBs']:.3f} pmc/moved {k.get('pmc_over_moved')}")
PY
for B in 0 1; do timeout 300 python tools/time_advect.py --size 256 --field tg --bc $B >> $O/time_advect.jsonl 2>> $O/adv.err; done
timeout 300 python tools/time_advect.py --size 384 --dtype f64 --field tg --bc 1 >> $O/time_advect.jsonl 2>> $O/adv.err
python - <<PY
import json
for l in open('$O/time_advect.jsonl'):
    d=json.loads(l); print(d['size'], d['dtype'], 'bc',d['bc'], d['ms_semi_lagrangian_staggered'])
PY
