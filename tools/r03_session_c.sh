#!/bin/bash
# Round-3 GPU visit C: FK refactor + lanes-as-waypoints collision kernel (tests + timing), FPS 1024-thread variant timing.
mkdir -p gpurun_out/r03c
export PYTHONUNBUFFERED=1
REPO=$(pwd); O=$REPO/gpurun_out/r03c
timeout 900 python -m pytest tests/test_gpu_franka.py tests/test_gpu_full_size.py tests/test_gpu_pointnet.py tests/test_gpu_rollout_success.py tests/test_gpu_metrics.py tests/test_gpu_loss.py tests/test_gpu_data.py -m gpu -q --tb=short -p no:cacheprovider > $O/pytest_gpu.log 2>&1
echo "pytest exit: $?" >> $O/pytest_gpu.log; grep -E "FAILED|ERROR" $O/pytest_gpu.log | head -20; tail -3 $O/pytest_gpu.log | cut -c1-300
python tools/collision_timing.py 8192 50 20 > $O/collision_c4.json 2> $O/collision_c4.err; cat $O/collision_c4.json
python tools/collision_timing.py 8192 70 20 > $O/collision_t70.json 2> $O/collision_t70.err; cat $O/collision_t70.json
python tools/fps_timing.py 8192 512 > $O/fps_timing.log 2>&1; cat $O/fps_timing.log
