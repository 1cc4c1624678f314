#!/bin/bash
# Three rocprofv3 passes (kernel trace + stats, FETCH_SIZE, WRITE_SIZE -- PMC passes separate, no other trace domains) per workload group
# of tools/path_workload.py, then tools/kernel_roofline.py -> one JSON table for every kernel on the path.
#   bash tools/kernel_roofline.sh gpurun_out/roofline [groups...]      (default groups: f32_256 f32_512 f64_384)
set -u
REPO="${GRAFT_REPO_ROOT:-$(cd "$(dirname "$0")/.." && pwd)}"
OUT="$(realpath -m "${1:-$REPO/gpurun_out/roofline}")"; shift || true
GROUPS_=("$@"); [ ${#GROUPS_[@]} -eq 0 ] && GROUPS_=(f32_256 f32_512 f64_384)
export TMPDIR=/tmp
mkdir -p "$OUT"
for G in "${GROUPS_[@]}"; do
  D="$OUT/$G"; rm -rf "$D"; mkdir -p "$D"
  # untraced: first-call autotune + wall time of one CG iteration; the traced passes pin these plans (no candidate shares a kernel name)
  (cd /tmp && timeout 600 python "$REPO/tools/path_workload.py" --group $G --write-plans "$D/plans.json" > "$D/plans.log" 2>&1); echo "$G plans rc=$?"
  (cd /tmp && timeout 600 rocprofv3 --kernel-trace --stats --output-format csv -d "$D/stats" -o k -- python "$REPO/tools/path_workload.py" --group $G --plans "$D/plans.json" --manifest "$D/manifest.json" > "$D/stats.log" 2>&1); echo "$G stats rc=$?"
  for CTR in FETCH_SIZE WRITE_SIZE; do
    (cd /tmp && timeout 600 rocprofv3 --kernel-trace --pmc $CTR --output-format csv -d "$D/$CTR" -o pmc -- python "$REPO/tools/path_workload.py" --group $G --plans "$D/plans.json" --reps 2 > "$D/$CTR.log" 2>&1); echo "$G $CTR rc=$?"
  done
done
python "$REPO/tools/kernel_roofline.py" $(for G in "${GROUPS_[@]}"; do echo "$OUT/$G"; done) > "$OUT/kernel_roofline.json"; echo "table rc=$?"
# the raw CSVs are large (one row per dispatch): keep the stats summaries and the table
find "$OUT" -name "*kernel_trace.csv" -size +2M -delete 2>/dev/null
find "$OUT" -name "*counter_collection.csv" -size +2M -delete 2>/dev/null
