#!/bin/bash
# Round-3 GPU session B: full suite, bench (quick), configs 5/4, f32_256 kernel table (new diffusion / obstacle kernels), CG1 sweep.
set -u
REPO="${GRAFT_REPO_ROOT:-/root/repo}"
cd "$REPO"
O=gpurun_out/r3b; mkdir -p $O
export TMPDIR=/tmp
STEPS="${1:-test,bench,configs,roofline,cg1}"
if [[ "$STEPS" == *test* ]]; then timeout 1800 python -m pytest tests -m gpu -q -p no:cacheprovider > $O/pytest_gpu.log 2>&1; echo "pytest rc=$?"; tail -8 $O/pytest_gpu.log; grep -n "closed 512\|config5 parity at" $O/pytest_gpu.log | head; fi
if [[ "$STEPS" == *bench* ]]; then timeout 600 python bench.py --steps 10 --warmup 2 --pmc 0 --cpu-size 0 > $O/bench_quick.json 2> $O/bench.err; echo "bench rc=$?"; python -c "
import json; d=json.load(open('$O/bench_quick.json')); print(d['value'], d['ms_per_step'], d['roofline']['frac'], d['config3']['ms_per_iteration'], d['plan'], d['config3']['plan'])"; fi
if [[ "$STEPS" == *configs* ]]; then timeout 600 python tools/bench_configs.py 5 4 > $O/configs.jsonl 2> $O/configs.err; echo "configs rc=$?"; cut -c1-700 $O/configs.jsonl; fi
if [[ "$STEPS" == *roofline* ]]; then bash tools/kernel_roofline.sh $O/roofline f32_256; python - <<PY
import json
d=json.load(open('$O/roofline/kernel_roofline.json'))
for g in d['groups']:
    for k in g['kernels']:
        if k.get('launches',0): print(f"{k['label'][:60]:60s} {k['avg_us']:8.1f} us frac {k['frac_of_8TBs']:.3f} pmc/moved {k.get('pmc_over_moved')}")
        else: print(k['label'][:60], 'NOT RUN')
PY
fi
if [[ "$STEPS" == *cg1* ]]; then timeout 600 python tools/sweep_cg1.py 200 > $O/cg1_sweep.jsonl 2> $O/cg1.err; echo "cg1 rc=$?"; cut -c1-300 $O/cg1_sweep.jsonl; fi
