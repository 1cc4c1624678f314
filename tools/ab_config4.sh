#!/bin/bash
# A/B of the first-call autotune on the latency-bound batched-smoke workload (bench.py --workload config4): analytic plan vs tuned
cd "${GRAFT_REPO_ROOT:-/root/repo}"
for B in 8 1; do
  for AT in 0 1; do
    PHIHIP_AUTOTUNE=$AT timeout 200 python bench.py --workload config4 --batch-total $B --steps 20 --warmup 3 --cg-iters 50 2>/dev/null > /tmp/ab.json
    python - "$B" "$AT" <<'PY'
import json, sys
d = json.load(open('/tmp/ab.json'))
print(json.dumps({"batch": int(sys.argv[1]), "autotune": int(sys.argv[2]), "ms_per_step": round(d["ms_per_step"], 4), "us_per_cg_iteration": d["us_per_cg_iteration_rank0"],
                  "plan": {k: [v["rows"], v["tpr"], v["nblk"]] for k, v in d["plan"].items()}}))
PY
  done
done
