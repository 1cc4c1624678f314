#!/bin/bash
# Round-3 GPU session J: SQ counters of the advection adjoints, global-atomic version (exp/libphihip_adjold.so) vs LDS windows (tree).
set -u
REPO="${GRAFT_REPO_ROOT:-/root/repo}"
cd "$REPO"
O=gpurun_out/r3j; mkdir -p $O
export TMPDIR=/tmp
P1="SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_INSTS_VMEM_WR"
P2="SQ_WAVE_CYCLES SQ_LDS_BANK_CONFLICT SQ_ACTIVE_INST_LDS SQ_WAIT_INST_LDS SQ_INSTS_VMEM_RD SQ_ACTIVE_INST_VMEM SQ_WAIT_ANY SQ_WAVES"
for V in new old; do
  LA=""; [ $V = old ] && LA="--lib $REPO/phiflow_amd/lib/exp/libphihip_adjold.so"
  for P in 1 2; do
    PM="$P1"; [ $P = 2 ] && PM="$P2"
    (cd /tmp && timeout 600 rocprofv3 --kernel-trace --pmc $PM --output-format csv -d "$REPO/$O/sq_${V}_$P" -o pmc -- python "$REPO/tools/path_workload.py" --group f32_256 --reps 2 $LA > "$REPO/$O/sq_${V}_$P.log" 2>&1); echo "sq $V $P rc=$?"
    python tools/sq_summary.py $O/sq_${V}_$P > $O/sq_${V}_$P.json
  done
done
python - <<PY
import json
for V in ('old','new'):
    for P in (1,2):
        try: d=json.load(open('$O/sq_%s_%d.json'%(V,P)))
        except Exception as e: print(V,P,e); continue
        for k,v in d.items():
            if 'bwd' in k and ('advect' in k or 'mac' in k): print(V,P,k[:48], {a:(round(b,3) if isinstance(b,float) and b<10 else int(b)) for a,b in v.items() if a!='launches'})
PY
