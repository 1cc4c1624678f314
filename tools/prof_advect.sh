#!/bin/bash
# rocprofv3 kernel stats + PMC passes of the advection kernels (tools/time_advect.py); outputs under gpurun_out/prof_adv*
REPO="${GRAFT_REPO_ROOT:-/root/repo}"; cd /tmp; export TMPDIR=/tmp
ARGS="${1:---size 256 --field tg}"
rocprofv3 --kernel-trace --stats --output-format csv -d "$REPO/gpurun_out/prof_adv_stats" -o adv -- python "$REPO/tools/time_advect.py" $ARGS --reps 5 > "$REPO/gpurun_out/prof_adv_stats.log" 2>&1
python - "$REPO" <<'PY'
import csv,glob,sys
repo=sys.argv[1]
for f in glob.glob(repo+'/gpurun_out/prof_adv_stats/**/*kernel_stats.csv', recursive=True):
    for r in csv.DictReader(open(f)):
        if 'advect' in r['Name']: print(r['Name'][:110], r['Calls'], r['AverageNs'])
PY
i=0
for CTRS in "SQ_WAVES SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR SQ_INSTS_SMEM SQ_INSTS_BRANCH" "SQ_BUSY_CYCLES SQ_WAVE_CYCLES SQ_WAIT_INST_ANY SQ_WAIT_INST_LDS SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_LDS SQ_ACTIVE_INST_VMEM SQ_ACTIVE_INST_SCA" "SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_LDS_ADDR_CONFLICT SQ_LDS_UNALIGNED_STALL SQ_INST_CYCLES_VMEM_RD SQ_INST_CYCLES_VMEM_WR SQ_IFETCH SQ_WAIT_ANY" "GRBM_GUI_ACTIVE TCC_HIT_sum TCC_MISS_sum TCP_TOTAL_CACHE_ACCESSES_sum"; do
  i=$((i+1))
  rocprofv3 --kernel-trace --pmc $CTRS --output-format csv -d "$REPO/gpurun_out/prof_adv_pmc/$i" -o pmc -- python "$REPO/tools/time_advect.py" $ARGS --reps 2 > "$REPO/gpurun_out/prof_adv_pmc_$i.log" 2>&1
done
python - "$REPO" <<'PY'
import csv,glob,sys,collections,json
repo=sys.argv[1]
acc=collections.defaultdict(lambda: collections.defaultdict(list))
for f in glob.glob(repo+'/gpurun_out/prof_adv_pmc/**/*counter_collection.csv', recursive=True):
    for r in csv.DictReader(open(f)):
        n=r['Kernel_Name']
        if 'advect' not in n: continue
        acc[n[:100]][r['Counter_Name']].append(float(r['Counter_Value']))
out={k:{c:sum(v)/len(v) for c,v in d.items()} for k,d in acc.items()}
print(json.dumps(out,indent=1))
json.dump(out,open(repo+'/gpurun_out/prof_adv_pmc.json','w'),indent=1)
PY
