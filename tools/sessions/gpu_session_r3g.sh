#!/bin/bash
# Round-3 GPU session G: advection after the wall split, marching diffusion, obstacle kernels; suite.
set -u
REPO="${GRAFT_REPO_ROOT:-/root/repo}"
cd "$REPO"
O=gpurun_out/r3g; mkdir -p $O
export TMPDIR=/tmp
STEPS="${1:-advect,roofline,configs,test}"
if [[ "$STEPS" == *advect* ]]; then
  for REP in 1 2; do for L in ab/libphihip_base.so ""; do
    LA=""; [ -n "$L" ] && LA="--lib $REPO/$L"
    timeout 300 python tools/time_advect.py --size 256 --field tg $LA >> $O/time_advect.jsonl 2>> $O/adv.err
    timeout 300 python tools/time_advect.py --size 256 --field tg --bc 1 $LA >> $O/time_advect.jsonl 2>> $O/adv.err
    timeout 300 python tools/time_advect.py --size 384 --dtype f64 --field tg --bc 1 $LA >> $O/time_advect.jsonl 2>> $O/adv.err
    timeout 300 python tools/time_advect.py --size 512 --field tg --bc 1 $LA >> $O/time_advect.jsonl 2>> $O/adv.err
  done; done
  python - <<PY
import json
for l in open('$O/time_advect.jsonl'):
    d=json.loads(l); print(d['lib'][:12], d['size'], d['dtype'], 'bc',d['bc'], d['ms_semi_lagrangian_staggered'], {k:v for k,v in d.items() if k.startswith('ms_semi_lagrangian_staggered_halo')})
PY
fi
if [[ "$STEPS" == *roofline* ]]; then bash tools/kernel_roofline.sh $O/roofline f32_256 f64_384; python - <<PY
import json
d=json.load(open('$O/roofline/kernel_roofline.json'))
for g in d['groups']:
    for k in g['kernels']:
        if k.get('launches',0): print(f"{g['group']} {k['label'][:58]:58s} {k['avg_us']:8.1f} us frac {k['frac_of_8TBs']:.3f} pmc/moved {k.get('pmc_over_moved')}")
PY
fi
if [[ "$STEPS" == *configs* ]]; then timeout 600 python tools/bench_configs.py 5 4 > $O/configs.jsonl 2> $O/configs.err; echo "configs rc=$?"; cut -c1-900 $O/configs.jsonl; fi
if [[ "$STEPS" == *test* ]]; then timeout 1800 python -m pytest tests -m gpu -q -p no:cacheprovider > $O/pytest_gpu.log 2>&1; echo "pytest rc=$?"; tail -4 $O/pytest_gpu.log; timeout 900 python tests/fuzz_parity.py --first 9000 --count 60 > $O/fuzz.log 2>&1; tail -1 $O/fuzz.log; fi
