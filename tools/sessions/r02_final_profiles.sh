#!/bin/bash
# round-2 evidence run: bench line (live PMC), rocprofv3 kernel stats of the bench step and of the 512^3 solve, PMC passes per size
REPO="${GRAFT_REPO_ROOT:-/root/repo}"; cd "$REPO"; mkdir -p gpurun_out; export TMPDIR=/tmp
timeout 400 python bench.py --steps 20 --warmup 5 > gpurun_out/r02_bench_n1.json 2> gpurun_out/r02_bench_n1.err; echo "bench rc=$?"
bash tools/prof_bench_stats.sh r02_prof_bench 0 | head -14
(cd /tmp && timeout 300 rocprofv3 --kernel-trace --stats --output-format csv -d "$REPO/gpurun_out/r02_prof_cg512" -o cg512 -- python "$REPO/tools/pmc_workload.py" 512 > "$REPO/gpurun_out/r02_prof_cg512.log" 2>&1); echo "stats512 rc=$?"
for SZ in 256 512; do
  for CTR in FETCH_SIZE WRITE_SIZE; do
    (cd /tmp && timeout 300 rocprofv3 --kernel-trace --pmc $CTR --output-format csv -d "$REPO/gpurun_out/r02_pmc_${SZ}/$CTR" -o pmc -- python "$REPO/tools/pmc_workload.py" $SZ > "$REPO/gpurun_out/r02_pmc_${SZ}_$CTR.log" 2>&1); echo "pmc $SZ $CTR rc=$?"
  done
  python tools/pmc_summary.py gpurun_out/r02_pmc_${SZ} gpurun_out/r02_pmc_summary_${SZ}.json > /dev/null
done
python tools/pmc_traffic.py gpurun_out/r02_pmc_summary_256.json gpurun_out/r02_pmc_summary_512.json > gpurun_out/r02_pmc_traffic.json; cat gpurun_out/r02_pmc_traffic.json | head -30
python - <<'PY'
import csv, glob
for f in glob.glob('gpurun_out/r02_prof_cg512/**/*kernel_stats.csv', recursive=True):
    rows = sorted(csv.DictReader(open(f)), key=lambda r: -float(r['TotalDurationNs']))
    for r in rows[:6]: print(r['Name'][:120], r['Calls'], round(float(r['AverageNs'])/1e3, 2))
PY
