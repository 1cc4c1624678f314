#!/bin/bash
# mid-size dip: streaming passes vs marching kernels per size, and the HBM traffic of the 384^3 kernels
REPO="${GRAFT_REPO_ROOT:-/root/repo}"; cd "$REPO"; mkdir -p gpurun_out; export TMPDIR=/tmp
timeout 300 python tools/mid_size_probe.py --sizes 224,256,288,320,384,448,512,640 > gpurun_out/r02_mid_size_probe.jsonl 2> gpurun_out/r02_mid_size_probe.err; echo "probe rc=$?"
cat gpurun_out/r02_mid_size_probe.jsonl
for CTR in FETCH_SIZE WRITE_SIZE; do
  (cd /tmp && timeout 300 rocprofv3 --kernel-trace --pmc $CTR --output-format csv -d "$REPO/gpurun_out/r02_pmc_384/$CTR" -o pmc -- python "$REPO/tools/pmc_workload.py" 384 > "$REPO/gpurun_out/r02_pmc_384_$CTR.log" 2>&1); echo "pmc 384 $CTR rc=$?"
done
python tools/pmc_summary.py gpurun_out/r02_pmc_384 gpurun_out/r02_pmc_summary_384.json > /dev/null
python - <<'PY'
import json
d=json.load(open('gpurun_out/r02_pmc_summary_384.json'))['kernels']
for k,v in d.items():
    if v.get('launches',0)>=15: print(k[:90], v['launches'], 'read MB', round(v['read_bytes_calibrated']/1e6,1), 'write MB', round(v['write_bytes_calibrated']/1e6,1))
PY
