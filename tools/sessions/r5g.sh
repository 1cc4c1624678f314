#!/bin/bash
# Round-5 GPU session G: the smoke256 workload as a same-box A/B against the round-4 library with a warm-up long enough for the adaptive reach to have
# settled in both builds (the one-off switch to the wide reach costs ~4.5 ms of first-call autotune: where it lands decided the r4 / r5 lines so far)
set -u
REPO="${GRAFT_REPO_ROOT:-/root/repo}"; cd "$REPO"
O=gpurun_out/${SESSION_TAG:-r5g}; mkdir -p $O; export TMPDIR=/tmp
for ROUND in 1 2; do for LIB in phiflow_amd/lib/libphihip_r4.so ""; do
  for W in 30 90; do
    PHIHIP_LIBRARY=$LIB timeout 300 python bench.py --workload smoke256 --steps 40 --warmup $W > $O/tmp.json 2>> $O/err.log
    python - <<PY >> $O/smoke256_ab.jsonl
import json
d=json.load(open('$O/tmp.json'))
print(json.dumps({"lib": "$LIB" or "HEAD", "warmup": $W, "steps": 40, "ms_per_step": round(d["ms_per_step"],4), "op_ms_profiled_step": d.get("op_ms_profiled_step"), "non_cg_share": d.get("non_cg_share_of_profiled_step"), "fallback": d.get("advect_fallback_last_call"), "build_id": d.get("build_id")}))
PY
  done
done; done
cat $O/smoke256_ab.jsonl | cut -c1-330
