#!/bin/bash
# hits-only ball-query rows: parity on the touched paths, then the bench line + kernel stats
mkdir -p gpurun_out/r03f
export PYTHONUNBUFFERED=1
REPO=$(pwd); O=$REPO/gpurun_out/r03f
timeout 900 python -m pytest tests/test_gpu_pointnet.py tests/test_gpu_shard.py -m gpu -q --tb=short -p no:cacheprovider > $O/pytest_gpu.log 2>&1; tail -4 $O/pytest_gpu.log | cut -c1-200
timeout 400 python bench.py --cpu-envs 0 --extra 0 --pipeline-steps 0 --all-slots-steps 0 > $O/bench.log 2>&1; tail -1 $O/bench.log | cut -c1-400
( cd /tmp && export TMPDIR=/tmp && rm -rf /tmp/st_h && timeout 600 rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/st_h -o run -- python $REPO/bench.py --steps 3 --warmup 1 --cpu-envs 0 --extra 0 --fast-steps 0 --pipeline-steps 0 --all-slots-steps 0 > $O/stats_head.log 2>&1
  find /tmp/st_h -name "*kernel_stats.csv" -exec cp {} $O/stats_head_kernel_stats.csv \; )
grep -E "ball_query|bq_sort" $O/stats_head_kernel_stats.csv | awk -F'",' '{print substr($1,1,60), $2}'
