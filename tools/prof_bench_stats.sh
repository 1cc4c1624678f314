#!/bin/bash
# rocprofv3 --kernel-trace --stats of the benchmark step (no CPU leg, no config-3 block, no nested PMC): per-kernel averages
REPO="${GRAFT_REPO_ROOT:-/root/repo}"; cd /tmp; export TMPDIR=/tmp
OUT="$REPO/gpurun_out/${1:-prof_bench}"
rocprofv3 --kernel-trace --stats --output-format csv -d "$OUT" -o bench -- python "$REPO/bench.py" --steps 5 --warmup 2 --cpu-size 0 --config3-size ${2:-0} --profile-steps 0 --pmc 0 --phi-level 0 > "$OUT.log" 2>&1
python - "$OUT" <<'PY'
import csv,glob,sys
for f in glob.glob(sys.argv[1]+'/**/*kernel_stats.csv', recursive=True):
    rows=list(csv.DictReader(open(f)))
    rows.sort(key=lambda r:-float(r['TotalDurationNs']))
    for r in rows[:24]: print(f"{r['Name'][:150]:150s} calls {r['Calls']:>6s} avg_us {float(r['AverageNs'])/1e3:9.2f} pct {r['Percentage']}")
PY
