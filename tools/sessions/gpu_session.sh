#!/bin/bash
# One GPU-box session: smoke, parity tests, bench, tile sweep, rocprofv3 kernel stats + PMC passes. Outputs under gpurun_out/.
set -u
REPO="${GRAFT_REPO_ROOT:-/root/repo}"
cd "$REPO"
mkdir -p gpurun_out
export TMPDIR=/tmp
STEPS="${1:-all}"
run() { echo "== $1"; shift; "$@"; echo "rc=$?"; }
if [[ "$STEPS" == *all* || "$STEPS" == *smoke* ]]; then
  echo "== smoke"; timeout 600 python -c "import __graft_entry__ as g; g.smoke()" > gpurun_out/smoke.log 2>&1; echo "rc=$?"; tail -2 gpurun_out/smoke.log
fi
if [[ "$STEPS" == *all* || "$STEPS" == *test* ]]; then
  echo "== pytest -m gpu"; timeout 1500 python -m pytest tests -m gpu -q -p no:cacheprovider > gpurun_out/pytest_gpu.log 2>&1; echo "rc=$?"; tail -25 gpurun_out/pytest_gpu.log
fi
if [[ "$STEPS" == *all* || "$STEPS" == *bench* ]]; then
  echo "== bench"; timeout 600 python bench.py --steps 10 --warmup 2 > gpurun_out/bench.log 2>&1; echo "rc=$?"; tail -2 gpurun_out/bench.log
fi
if [[ "$STEPS" == *all* || "$STEPS" == *sweep* ]]; then
  echo "== sweep 256"; timeout 300 python tools/sweep_cg.py --size 256 --iters 10 > gpurun_out/sweep_256.jsonl 2> gpurun_out/sweep_256.err; echo "rc=$?"
  echo "== sweep 512"; timeout 300 python tools/sweep_cg.py --size 512 --iters 6 > gpurun_out/sweep_512.jsonl 2> gpurun_out/sweep_512.err; echo "rc=$?"
fi
if [[ "$STEPS" == *all* || "$STEPS" == *configs* ]]; then
  echo "== configs 3/4/5"; timeout 600 python tools/bench_configs.py > gpurun_out/configs.jsonl 2> gpurun_out/configs.err; echo "rc=$?"; cat gpurun_out/configs.jsonl; tail -3 gpurun_out/configs.err
fi
if [[ "$STEPS" == *all* || "$STEPS" == *prof* ]]; then
  echo "== rocprofv3 stats (bench)"
  (cd /tmp && timeout 600 rocprofv3 --kernel-trace --stats --output-format csv -d "$REPO/gpurun_out/prof_stats" -o bench -- python "$REPO/bench.py" --steps 3 --warmup 1 --cpu-size 0 --profile-steps 0 > "$REPO/gpurun_out/rocprof_bench.log" 2>&1); echo "rc=$?"
  echo "== rocprofv3 stats (512^3 CG)"
  (cd /tmp && timeout 600 rocprofv3 --kernel-trace --stats --output-format csv -d "$REPO/gpurun_out/prof_stats_cg512" -o cg512 -- python "$REPO/tools/pmc_workload.py" 512 > "$REPO/gpurun_out/rocprof_cg512.log" 2>&1); echo "rc=$?"
  for SZ in 256 512; do
    for CTR in FETCH_SIZE WRITE_SIZE; do
      echo "== rocprofv3 pmc $CTR ($SZ^3)"
      (cd /tmp && timeout 600 rocprofv3 --kernel-trace --pmc $CTR --output-format csv -d "$REPO/gpurun_out/pmc_${SZ}/$CTR" -o pmc -- python "$REPO/tools/pmc_workload.py" $SZ > "$REPO/gpurun_out/pmc_${SZ}_$CTR.log" 2>&1); echo "rc=$?"
    done
    python tools/pmc_summary.py gpurun_out/pmc_${SZ} gpurun_out/pmc_summary_${SZ}.json > /dev/null
  done
  python tools/pmc_traffic.py gpurun_out/pmc_summary_256.json gpurun_out/pmc_summary_512.json > gpurun_out/pmc_traffic.json; cat gpurun_out/pmc_traffic.json
  find gpurun_out -name "*.csv" | head -30
fi
