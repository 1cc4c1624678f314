#!/bin/bash
# Round-3 GPU session D: family sweeps of the config-5 CG kernels (fp64 + flags, 384^3) and of 384^3 / 512^3 fp32; configs; bench; suite.
set -u
REPO="${GRAFT_REPO_ROOT:-/root/repo}"
cd "$REPO"
O=gpurun_out/r3d; mkdir -p $O
export TMPDIR=/tmp
STEPS="${1:-sweep,configs,bench,test}"
CFG="1,16,16;1,16,32;1,16,64;2,16,32;2,16,64;2,32,16;2,32,32;2,32,64;2,32,128;4,32,32;4,32,64;4,64,32;4,64,64;4,64,128;1,64,16;1,64,32;1,64,64;1,64,128;2,64,16;2,64,32;2,64,64;2,64,128;1,32,16;1,32,32;1,32,64;1,32,128"
if [[ "$STEPS" == *sweep* ]]; then
  for F in 1 2 3; do timeout 300 python tools/sweep_cg.py --size 384 --dtype f64 --obstacle 1 --iters 20 --family $F --configs "$CFG" > $O/sweep_f64_384_flags_fam$F.jsonl 2>> $O/sweep.err; echo "sweep f64 fam $F rc=$?"; done
  for F in 1 3; do timeout 300 python tools/sweep_cg.py --size 384 --iters 20 --family $F --configs "$CFG" > $O/sweep_f32_384_fam$F.jsonl 2>> $O/sweep.err; echo "sweep f32 384 fam $F rc=$?"; done
  timeout 300 python tools/sweep_cg.py --size 512 --iters 12 --family 1 --configs "$CFG" > $O/sweep_f32_512_fam1.jsonl 2>> $O/sweep.err; echo "sweep f32 512 rc=$?"
  python - <<PY
import json,glob
for f in sorted(glob.glob('$O/sweep_*.jsonl')):
    rows=[json.loads(l) for l in open(f)]
    fam=rows[0]['family']; key={1:'ms_matvec',2:'ms_update',3:'ms_update_r'}[fam]
    rows.sort(key=lambda r:r[key])
    print(f.split('/')[-1], 'auto:', [ (r[key], r['plan_mv' if fam==1 else ('plan_up' if fam==2 else 'plan_ur')][:3]) for r in rows if r['rows']==0])
    for r in rows[:6]: print('   ', r['rows'], r['tpr'], r['chunk'], r[key])
PY
fi
if [[ "$STEPS" == *configs* ]]; then for REP in 1 2; do timeout 600 python tools/bench_configs.py 5 >> $O/configs.jsonl 2>> $O/configs.err; done; timeout 300 python tools/bench_configs.py 4 3 >> $O/configs.jsonl 2>> $O/configs.err; echo "configs rc=$?"; cut -c1-900 $O/configs.jsonl; fi
if [[ "$STEPS" == *bench* ]]; then timeout 600 python bench.py --steps 20 --warmup 5 --pmc 0 --cpu-size 0 > $O/bench_quick.json 2> $O/bench.err; echo "bench rc=$?"; python -c "
import json; d=json.load(open('$O/bench_quick.json')); print(d['value'], d['ms_per_step'], d['roofline']['frac'], d['config3']['ms_per_iteration'], d['plan'], d['config3']['plan'])"; fi
if [[ "$STEPS" == *test* ]]; then timeout 1800 python -m pytest tests -m gpu -q -p no:cacheprovider -s > $O/pytest_gpu.log 2>&1; echo "pytest rc=$?"; tail -4 $O/pytest_gpu.log; grep -n "closed 512\|config5 parity at\|config2 parity\|config3 parity" $O/pytest_gpu.log | cut -c1-400 | head; fi
