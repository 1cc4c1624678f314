#!/bin/bash
# Round-4 GPU session N: the mid-size dip seen from the L2's memory side -- per-channel read requests, DRAM credit stalls and queue level of the
# CG kernels at 384^3 against 512^3 and 256^3 (separate PMC passes, kernel trace only)
set -u
REPO="${GRAFT_REPO_ROOT:-/root/repo}"
cd "$REPO"
O=gpurun_out/r4n; mkdir -p $O
export TMPDIR=/tmp
for N in 384 512 320; do
  for SET in "TCC_EA0_RDREQ" "TCC_EA0_RDREQ_DRAM_CREDIT_STALL_sum TCC_EA0_RDREQ_LEVEL_sum TCC_BUSY_avr GRBM_GUI_ACTIVE" "TCC_REQ_sum TCC_READ_sum TCC_HIT_sum TCC_MISS_sum"; do
    TAG=$(echo $SET | cut -d' ' -f1)
    (cd /tmp && PHIHIP_AUTOTUNE=0 timeout 300 rocprofv3 --kernel-trace --pmc $SET --output-format csv -d "$REPO/$O/n${N}_$TAG" -o pmc -- python "$REPO/tools/pmc_workload.py" $N > "$REPO/$O/n${N}_$TAG.log" 2>&1); echo "$N $TAG rc=$?"
  done
done
python - <<'PY'
import csv, glob, os, json, collections
csv.field_size_limit(1 << 30)
O = "gpurun_out/r4n"
out = {}
for d in sorted(glob.glob(O + "/n*_*")):
    if not os.path.isdir(d): continue
    per = collections.defaultdict(lambda: collections.defaultdict(list))
    for f in glob.glob(d + "/**/*counter_collection.csv", recursive=True):
        rows = list(csv.DictReader(open(f)))
        if rows and os.path.basename(d).endswith("TCC_EA0_RDREQ"):
            print(os.path.basename(d), "columns:", list(rows[0].keys()))
        for r in rows:
            k = r["Kernel_Name"]
            if "march_kernel" not in k: continue
            per[k[:70]][r["Counter_Name"]].append(float(r["Counter_Value"]))
    out[os.path.basename(d)] = {k: {c: {"n": len(v), "mean": sum(v) / len(v), "min": min(v), "max": max(v)} for c, v in cs.items()} for k, cs in per.items()}
json.dump(out, open(O + "/tcc_channels.json", "w"), indent=1)
for d, ks in out.items():
    for k, cs in ks.items():
        print(d, k[20:70], {c: (round(s["mean"], 1), round(s["min"], 1), round(s["max"], 1), s["n"]) for c, s in cs.items()})
PY
find $O -name "*kernel_trace.csv" -size +2M -delete 2>/dev/null
find $O -name "*counter_collection.csv" -size +3M -delete 2>/dev/null
