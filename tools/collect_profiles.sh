#!/bin/bash
# copies the records of one tools/sessions/r5_final.sh run (gpurun_out/<tag>) into profiles/ under the names DESIGN.md / README.md / profiles/README.md cite
#   bash tools/collect_profiles.sh r5z
set -u
cd "$(dirname "$0")/.."
O=gpurun_out/${1:-r5z}; P=profiles; R=r05
cp $O/bench_n1.json $P/${R}_bench_n1.json; cp $O/bench_smoke256.json $P/${R}_bench_smoke256.json
cp $O/bench_config4.json $P/${R}_bench_config4.json; cp $O/bench_config4_resident.json $P/${R}_bench_config4_resident.json
cp $O/roofline/kernel_roofline.json $P/${R}_kernel_roofline.json
for g in f32_256 f32_512 f64_384; do f=$(find $O/roofline/$g/stats -name "*kernel_stats.csv" | head -1); [ -n "$f" ] && cp $f $P/${R}_kernel_stats_$g.csv; done
f=$(find $O/prof_bench -name "*kernel_stats.csv" | head -1); [ -n "$f" ] && cp $f $P/${R}_bench256_kernel_stats.csv
cp $O/configs_345.jsonl $P/${R}_configs_345.jsonl; cp $O/time_frow.jsonl $P/${R}_time_frow_final.jsonl
cp $O/smoke256_ab.jsonl $P/${R}_smoke256_ab.jsonl; cp $O/issue_rates.txt $P/${R}_issue_rates.txt; cp $O/backward_step.jsonl $P/${R}_backward_step.jsonl; cp $O/host_api.jsonl $P/${R}_host_api.jsonl
(cat $O/build_id.txt; grep -E "passed|failed" $O/pytest_gpu.log | tail -1; tail -1 $O/smoke.log) > $P/${R}_gpu_suite_final.txt
(cat $O/build_id.txt; echo "$(grep -c '^ok' $O/fuzz.log) randomised cases ok, $(grep -c '^skip' $O/fuzz.log) skipped (tests/fuzz_parity.py --first 53000 --count 120, resident-solver arm on every 2-D case):"; tail -1 $O/fuzz.log; grep "^FAIL" $O/fuzz.log) > $P/${R}_fuzz_gpu_final.txt
cat $P/${R}_gpu_suite_final.txt $P/${R}_fuzz_gpu_final.txt
