#!/bin/bash
# Round-3 GPU session F: the numbers DESIGN.md cites -- full bench line, per-kernel roofline table (three groups), configs 3-5, rocprofv3 stats of the bench.
set -u
REPO="${GRAFT_REPO_ROOT:-/root/repo}"
cd "$REPO"
O=gpurun_out/r3f; mkdir -p $O
export TMPDIR=/tmp
STEPS="${1:-bench,configs,roofline,stats}"
if [[ "$STEPS" == *bench* ]]; then timeout 900 python bench.py --steps 20 --warmup 5 > $O/bench.json 2> $O/bench.err; echo "bench rc=$?"; head -c 400 $O/bench.json; echo; fi
if [[ "$STEPS" == *configs* ]]; then timeout 600 python tools/bench_configs.py 3 4 5 > $O/configs.jsonl 2> $O/configs.err; echo "configs rc=$?"; cut -c1-400 $O/configs.jsonl; timeout 300 python bench.py --workload config4 --steps 20 --warmup 5 --cg-iters 50 > $O/bench_config4.json 2>> $O/bench.err; timeout 300 python bench.py --workload slab --size 256 --steps 5 --warmup 2 > $O/bench_slab256.json 2>> $O/bench.err; echo "config4/slab rc=$?"; head -c 300 $O/bench_slab256.json; echo; fi
if [[ "$STEPS" == *roofline* ]]; then bash tools/kernel_roofline.sh $O/roofline; fi
if [[ "$STEPS" == *stats* ]]; then
  (cd /tmp && timeout 600 rocprofv3 --kernel-trace --stats --output-format csv -d "$REPO/$O/prof_bench" -o bench -- python "$REPO/bench.py" --steps 3 --warmup 1 --cpu-size 0 --profile-steps 0 --pmc 0 > "$REPO/$O/rocprof_bench.log" 2>&1); echo "stats rc=$?"
  find $O/prof_bench -name "*kernel_trace.csv" -size +2M -delete
fi
