#!/bin/bash
# Round-3 GPU session E: advection A/B (round-2 library vs the CONSTS split), configs, bench, full suite.
set -u
REPO="${GRAFT_REPO_ROOT:-/root/repo}"
cd "$REPO"
O=gpurun_out/r3e; mkdir -p $O
export TMPDIR=/tmp
STEPS="${1:-advect,configs,bench,test}"
if [[ "$STEPS" == *advect* ]]; then
  for REP in 1 2; do for L in ab/libphihip_base.so ""; do
    LA=""; [ -n "$L" ] && LA="--lib $REPO/$L"
    timeout 300 python tools/time_advect.py --size 256 --field tg $LA >> $O/time_advect.jsonl 2>> $O/adv.err
    timeout 300 python tools/time_advect.py --size 512 --field tg $LA >> $O/time_advect.jsonl 2>> $O/adv.err
    timeout 300 python tools/time_advect.py --size 256 --field tg --bc 1 $LA >> $O/time_advect.jsonl 2>> $O/adv.err
    timeout 300 python tools/time_advect.py --size 384 --dtype f64 --field tg --bc 1 $LA >> $O/time_advect.jsonl 2>> $O/adv.err
    timeout 300 python tools/time_advect.py --size 256 --dtype f64 --field tg $LA >> $O/time_advect.jsonl 2>> $O/adv.err
  done; done
  echo "advect rc=$?"; cut -c1-420 $O/time_advect.jsonl
fi
if [[ "$STEPS" == *configs* ]]; then timeout 600 python tools/bench_configs.py 5 >> $O/configs.jsonl 2>> $O/configs.err; echo "configs rc=$?"; cut -c1-900 $O/configs.jsonl; fi
if [[ "$STEPS" == *bench* ]]; then timeout 600 python bench.py --steps 20 --warmup 5 --pmc 0 --cpu-size 0 > $O/bench_quick.json 2> $O/bench.err; echo "bench rc=$?"; python -c "
import json; d=json.load(open('$O/bench_quick.json')); print(d['value'], d['ms_per_step'], d['roofline']['frac'], d['config3']['ms_per_iteration'], d['kernel_ms_per_step'])"; fi
if [[ "$STEPS" == *test* ]]; then timeout 1800 python -m pytest tests -m gpu -q -p no:cacheprovider > $O/pytest_gpu.log 2>&1; echo "pytest rc=$?"; tail -4 $O/pytest_gpu.log; fi
