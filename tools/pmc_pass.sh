#!/bin/bash
# usage: tools/pmc_pass.sh <tag> "<counters>" <python script + args>   -> gpurun_out/pmc_<tag>.csv (per kernel, per counter)
# (PMC_TIMEOUT: seconds per pass, default 600 -- keep it SHORT for untried counter groups: a pass with the derived TCC_*_sum
# counters did not finish on this pool and cost 15 GPU-minutes)
tag=$1; ctr=$2; shift 2
REPO=$(pwd); mkdir -p gpurun_out
( cd /tmp && export TMPDIR=/tmp && rm -rf /tmp/pmc_$tag && timeout ${PMC_TIMEOUT:-600} rocprofv3 --pmc $ctr --kernel-trace --output-format csv -d /tmp/pmc_$tag -o run -- python $REPO/"$@" > $REPO/gpurun_out/pmc_${tag}.log 2>&1
  for f in $(find /tmp/pmc_$tag -name "*counter_collection.csv"); do python $REPO/tools/pmc_summary.py $f > $REPO/gpurun_out/pmc_${tag}.csv; done )
grep -E "bf16_kernel|packed_kernel|resident_kernel|pairs_kernel|fps_" gpurun_out/pmc_${tag}.csv | cut -c1-40,90- | head -40
