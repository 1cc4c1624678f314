#!/bin/bash
# copies the records of one tools/gpu_session.sh run (gpurun_out/<tag>) into profiles/ under the r06_ names DESIGN.md / README.md / profiles/README.md cite
#   bash tools/collect_profiles.sh r6z
set -u
cd "$(dirname "$0")/.."
O=gpurun_out/${1:?tag}; P=profiles; R=r06
c() { [ -f "$1" ] && cp "$1" "$2" && echo "  $2"; }
c $O/bench_n1.json $P/${R}_bench_n1.json
c $O/smoke256_w30.json $P/${R}_bench_smoke256.json
for r in 0 -1 2; do c $O/config4_res$r.json $P/${R}_bench_config4_resident$r.json; done
c $O/roofline/kernel_roofline.json $P/${R}_kernel_roofline.json
for g in f32_256 f32_512 f64_384; do f=$(find $O/roofline/$g/stats -name "*kernel_stats.csv" 2>/dev/null | head -1); [ -n "$f" ] && c $f $P/${R}_kernel_stats_$g.csv; done
f=$(find $O/prof_bench -name "*kernel_stats.csv" 2>/dev/null | head -1); [ -n "$f" ] && c $f $P/${R}_bench256_kernel_stats.csv
c $O/configs.jsonl $P/${R}_configs_3456.jsonl
for f in $O/time_frow_*.jsonl; do [ -f "$f" ] && c $f $P/${R}_final_$(basename $f); done
c $O/backward_step.jsonl $P/${R}_backward_step.jsonl
for f in $O/sweep_resident_*.jsonl; do [ -f "$f" ] && c $f $P/${R}_$(basename $f); done
c $O/jit_foreach_debug.txt $P/${R}_jit_foreach_debug_final.txt
if [ -f $O/pytest_tests.log ] || [ -f $O/pytest_tests_.log ]; then
  L=$(ls $O/pytest_tests*.log | head -1)
  (cat $O/build_id.txt; grep -E "passed|failed" $L | tail -1; tail -1 $O/smoke.log) > $P/${R}_gpu_suite_final.txt; echo "  $P/${R}_gpu_suite_final.txt"; cat $P/${R}_gpu_suite_final.txt
fi
if [ -f $O/fuzz.txt ]; then (cat $O/build_id.txt; echo "$(grep -c '^ok' $O/fuzz.txt) randomised cases ok, $(grep -c '^skip' $O/fuzz.txt) skipped (tests/fuzz_parity.py, resident-solver arm on every 2-D case):"; tail -1 $O/fuzz.txt; grep "^FAIL" $O/fuzz.txt) > $P/${R}_fuzz_gpu_final.txt; echo "  $P/${R}_fuzz_gpu_final.txt"; cat $P/${R}_fuzz_gpu_final.txt; fi
