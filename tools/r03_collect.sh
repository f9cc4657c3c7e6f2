#!/bin/bash
# gpurun_out/r03ev/ (tools/r03_evidence.sh) -> profiles/r03_* (tracked).
set -e
S=gpurun_out/r03ev; P=profiles
tail -1 $S/bench.log > $P/r03_bench_n1.json
cp $S/stats_head_kernel_stats.csv $P/r03_bench_kernel_stats.csv
for i in 1 2 3 4 5; do cp $S/pmc_head$i.csv $P/r03_pmc_pass${i}_envs8192.csv; done
T=$(mktemp -d); for i in 1 2 3; do cp $S/pmc_head$i.csv $T/pass${i}_summary.csv; done
python tools/pmc_traffic.py $T 8192 $P/r03_traffic.json > /dev/null
for i in 1 2 3 4 5; do cp $S/pmc_fast$i.csv $P/r03_fast_pmc_pass$i.csv; done
cp $S/stats_fast_kernel_stats.csv $P/r03_fast_mode_kernel_stats.csv
python tools/pmc_fast.py $S 8192 $P/r03_fast_traffic.json > /dev/null
for i in 1 2 3; do cp $S/pmc_col$i.csv $P/r03_collision_pmc_pass$i.csv; done
cp $S/stats_col_kernel_stats.csv $P/r03_collision_kernel_stats.csv
cp $S/collision_c4.json $P/r03_collision_c4.json; cp $S/collision_c2.json $P/r03_collision_c2.json
cp $S/pytest_gpu.log $P/r03_pytest_gpu.log
cp $S/soak_hashes.json $P/r03_soak_hashes.json; cp $S/horizon_report.json $P/r03_horizon_report.json
cp $S/sa3_phase_probe.log $P/r03_sa3_phase_probe.log
cp $S/box.log $P/r03_box.log
ls -la $P/r03_* | awk '{print $5, $9}'
