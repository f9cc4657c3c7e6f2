#!/bin/bash
# Round-3 GPU visit D: fused group-all kernel (tests + headline timing + kernel stats), collision chunk-size probes.
mkdir -p gpurun_out/r03d
export PYTHONUNBUFFERED=1
REPO=$(pwd); O=$REPO/gpurun_out/r03d
timeout 1200 python -m pytest tests/test_gpu_policy.py tests/test_gpu_shard.py tests/test_gpu_horizon.py tests/test_gpu_full_size.py tests/test_gpu_franka.py -m gpu -q --tb=short -p no:cacheprovider > $O/pytest_gpu.log 2>&1
echo "pytest exit: $?" >> $O/pytest_gpu.log; grep -E "FAILED|ERROR|Error" $O/pytest_gpu.log | head -20; tail -3 $O/pytest_gpu.log | cut -c1-300
timeout 600 python bench.py --steps 4 --warmup 1 --cpu-envs 0 --extra 0 --fast-steps 0 --pipeline-steps 0 --all-slots-steps 0 > $O/bench_short.log 2> $O/bench_short.err; python - <<'PY'
import json
d=json.loads(open('gpurun_out/r03d/bench_short.log').read().strip().splitlines()[-1])
print(d['value'], d['ms_per_step'], d['kernels_ms'])
PY
( cd /tmp && export TMPDIR=/tmp && rm -rf /tmp/st_h && timeout 600 rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/st_h -o run -- python $REPO/bench.py --steps 3 --warmup 1 --cpu-envs 0 --extra 0 --fast-steps 0 --pipeline-steps 0 --all-slots-steps 0 > $O/stats_head.log 2>&1
  find /tmp/st_h -name "*kernel_stats.csv" -exec cp {} $O/stats_head_kernel_stats.csv \; )
head -8 $O/stats_head_kernel_stats.csv | cut -c1-60,150-330
for tc in 0 32 16 264 164; do MPX_COL_TC_PROBE=$tc python tools/collision_timing.py 8192 50 20 2>/dev/null | python -c "import sys,json; d=json.loads(sys.stdin.read()); print('tc probe $tc:', d['ms'], d['frac_of_valu_peak'])"; done
