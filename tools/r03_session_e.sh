#!/bin/bash
mkdir -p gpurun_out/r03e
export PYTHONUNBUFFERED=1
REPO=$(pwd); O=$REPO/gpurun_out/r03e
timeout 600 python -m pytest tests/test_gpu_policy.py -m gpu -q --tb=short -p no:cacheprovider -k "chain or threshold" > $O/pytest_gpu.log 2>&1; tail -3 $O/pytest_gpu.log | cut -c1-200
( cd /tmp && export TMPDIR=/tmp && rm -rf /tmp/st_h && timeout 600 rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/st_h -o run -- python $REPO/bench.py --steps 3 --warmup 1 --cpu-envs 0 --extra 0 --fast-steps 0 --pipeline-steps 0 --all-slots-steps 0 > $O/stats_head.log 2>&1
  find /tmp/st_h -name "*kernel_stats.csv" -exec cp {} $O/stats_head_kernel_stats.csv \; )
grep -E "sa3_chain|packed_kernel" $O/stats_head_kernel_stats.csv | awk -F'",' '{print substr($1,1,50), $2}'
tail -1 $O/stats_head.log | cut -c1-200
