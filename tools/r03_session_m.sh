#!/bin/bash
mkdir -p gpurun_out/r03m
export PYTHONUNBUFFERED=1
REPO=$(pwd); O=$REPO/gpurun_out/r03m
timeout 900 python -m pytest tests/test_gpu_policy.py tests/test_gpu_training.py tests/test_gpu_edge_cases.py -m gpu -q --tb=short -p no:cacheprovider > $O/pytest_gpu.log 2>&1; tail -4 $O/pytest_gpu.log | cut -c1-200
python tools/linear_calls.py 8192 2>&1 | grep -E "68, 4194304|total"
MPX_LIB_PATH=$REPO/build_ab/libmpinets_hip_rl2.so python tools/linear_calls.py 8192 2>&1 | grep -E "68, 4194304|total"
