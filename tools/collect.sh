#!/bin/bash
# gpurun_out/<round>ev/ (tools/evidence.sh <round>) -> profiles/<round>_* (tracked).  usage: bash tools/collect.sh r05
set -e
R=${1:-r06}
S=gpurun_out/${R}ev; P=profiles
tail -1 $S/bench.log > $P/${R}_bench_n1.json
cp $S/stats_head_kernel_stats.csv $P/${R}_bench_kernel_stats.csv
for i in 1 2 3 4 5; do cp $S/pmc_head$i.csv $P/${R}_pmc_pass${i}_envs8192.csv; done
T=$(mktemp -d); for i in 1 2 3; do cp $S/pmc_head$i.csv $T/pass${i}_summary.csv; done
python tools/pmc_traffic.py $T 8192 $P/${R}_traffic.json > /dev/null
for i in 1 2 3 4 5; do cp $S/pmc_fast$i.csv $P/${R}_fast_pmc_pass$i.csv; done
cp $S/stats_fast_kernel_stats.csv $P/${R}_fast_mode_kernel_stats.csv
python tools/pmc_fast.py $S 8192 $P/${R}_fast_traffic.json > /dev/null
for i in 1 2 3; do cp $S/pmc_col$i.csv $P/${R}_collision_pmc_pass$i.csv; done
cp $S/stats_col_kernel_stats.csv $P/${R}_collision_kernel_stats.csv
cp $S/collision_c4.json $P/${R}_collision_c4.json; cp $S/collision_c2.json $P/${R}_collision_c2.json
[ -f $S/pytest_gpu.log ] && cp $S/pytest_gpu.log $P/${R}_pytest_gpu.log
cp $S/soak_hashes.json $P/${R}_soak_hashes.json; cp $S/horizon_report.json $P/${R}_horizon_report.json
cp $S/stats_train_kernel_stats.csv $P/${R}_train_step_kernel_stats.csv; cat $S/train_10.log $S/train_256.log > $P/${R}_train_step.log
cp $S/stats_trainx3_kernel_stats.csv $P/${R}_train_step_bf16x3_kernel_stats.csv
grep -v amdgpu.ids $S/sa2_bf16_phase_probe.log > $P/${R}_sa2_bf16_phase_probe.log
cp $S/box.log $P/${R}_box.log
ls -la $P/${R}_* | awk '{print $5, $9}'
grep -v amdgpu.ids $S/sa3_front_timing.log > $P/${R}_sa3_front_timing.log
cp $S/mfma_power_probe.log $P/${R}_mfma_power_probe.log
