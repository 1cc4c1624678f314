#!/bin/bash
# copies the records of one tools/sessions/gpu_session_r4k.sh run (gpurun_out/<tag>) into profiles/ under the names DESIGN.md / README.md cite
#   bash tools/collect_profiles.sh r4m
set -u
cd "$(dirname "$0")/.."
O=gpurun_out/${1:-r4k}; P=profiles
cp $O/bench_n1.json $P/r04_bench_n1.json; cp $O/bench_smoke256.json $P/r04_bench_smoke256.json
cp $O/bench_config4.json $P/r04_bench_config4.json; cp $O/bench_config4_resident.json $P/r04_bench_config4_resident.json
cp $O/roofline/kernel_roofline.json $P/r04_kernel_roofline.json
for g in f32_256 f32_512 f64_384; do f=$(find $O/roofline/$g/stats -name "*kernel_stats.csv" | head -1); [ -n "$f" ] && cp $f $P/r04_kernel_stats_$g.csv; done
f=$(find $O/prof_bench -name "*kernel_stats.csv" | head -1); [ -n "$f" ] && cp $f $P/r04_bench256_kernel_stats.csv
cp $O/configs_345.jsonl $P/r04_configs_345.jsonl; cp $O/time_frow.jsonl $P/r04_time_frow_final.jsonl; cp $O/sweep_resident.jsonl $P/r04_sweep_resident_final.jsonl
(cat $O/build_id.txt; grep -E "passed|failed" $O/pytest_gpu.log | tail -1; tail -1 $O/smoke.log; tail -1 $O/fuzz.log) > $P/r04_gpu_suite_final.txt
cp $O/tcc_384.json $P/r04_tcc_384.json; cp $O/tcc_512.json $P/r04_tcc_512.json
cat $P/r04_gpu_suite_final.txt
