#!/bin/bash
# Round-3 GPU session A: suite, bench line, config-5 A/B (gradient kernel), per-kernel roofline table (+ SQ counters for the CG kernels).
set -u
REPO="${GRAFT_REPO_ROOT:-/root/repo}"
cd "$REPO"
O=gpurun_out/r3a; mkdir -p $O
export TMPDIR=/tmp
STEPS="${1:-smoke,test,bench,configs,roofline,sq}"
if [[ "$STEPS" == *smoke* ]]; then timeout 600 python -c "import __graft_entry__ as g; g.smoke()" > $O/smoke.log 2>&1; echo "smoke rc=$?"; tail -1 $O/smoke.log; fi
if [[ "$STEPS" == *test* ]]; then timeout 1500 python -m pytest tests -m gpu -q -p no:cacheprovider -x > $O/pytest_gpu.log 2>&1; echo "pytest rc=$?"; tail -5 $O/pytest_gpu.log; fi
if [[ "$STEPS" == *bench* ]]; then timeout 900 python bench.py --steps 10 --warmup 2 > $O/bench.json 2> $O/bench.err; echo "bench rc=$?"; head -c 600 $O/bench.json; echo; tail -3 $O/bench.err; fi
if [[ "$STEPS" == *configs* ]]; then
  for L in base new; do
    if [ $L = base ]; then export PHIHIP_LIBRARY=$REPO/ab/libphihip_base.so; else unset PHIHIP_LIBRARY; fi
    for REP in 1; do timeout 600 python tools/bench_configs.py 5 3 >> $O/configs_$L.jsonl 2>> $O/configs.err; echo "configs $L rc=$?"; done
  done
  unset PHIHIP_LIBRARY
  cat $O/configs_base.jsonl $O/configs_new.jsonl | cut -c1-900
fi
if [[ "$STEPS" == *roofline* ]]; then bash tools/kernel_roofline.sh $O/roofline; fi
if [[ "$STEPS" == *sq* ]]; then
  for G in f64_384 f32_512; do
    (cd /tmp && timeout 600 rocprofv3 --kernel-trace --pmc SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_VALU SQ_INSTS_VALU SQ_INSTS_LDS --output-format csv -d "$REPO/$O/sq_$G" -o pmc -- python "$REPO/tools/path_workload.py" --group $G --reps 2 > "$REPO/$O/sq_$G.log" 2>&1); echo "sq $G rc=$?"
    python tools/sq_summary.py $O/sq_$G > $O/sq_$G.json; head -c 1500 $O/sq_$G.json; echo
    find $O/sq_$G -name "*.csv" -size +2M -delete
  done
fi
