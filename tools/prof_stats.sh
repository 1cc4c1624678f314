#!/bin/bash
# rocprofv3 --kernel-trace --stats of any command: bash tools/prof_stats.sh OUTDIR command...   (per-kernel averages, top 30 by total time)
REPO="${GRAFT_REPO_ROOT:-/root/repo}"; OUT="$REPO/$1"; shift
mkdir -p "$OUT"; cd /tmp; export TMPDIR=/tmp
( cd "$REPO" && rocprofv3 --kernel-trace --stats --output-format csv -d "$OUT" -o run -- "$@" ) > "$OUT.log" 2>&1
python - "$OUT" <<'PY'
import csv,glob,sys
for f in glob.glob(sys.argv[1]+'/**/*kernel_stats.csv', recursive=True):
    rows=list(csv.DictReader(open(f)))
    rows.sort(key=lambda r:-float(r['TotalDurationNs']))
    for r in rows[:30]: print(f"{r['Name'][:130]:130s} calls {r['Calls']:>6s} avg_us {float(r['AverageNs'])/1e3:9.2f} pct {r['Percentage']}")
PY
