#!/bin/bash
mkdir -p gpurun_out/r03n
export PYTHONUNBUFFERED=1
REPO=$(pwd); O=$REPO/gpurun_out/r03n
timeout 600 python -m pytest tests/test_gpu_pointnet.py tests/test_gpu_soak.py -m gpu -q --tb=short -p no:cacheprovider > $O/pytest_gpu.log 2>&1; tail -3 $O/pytest_gpu.log | cut -c1-200
bash tools/r03_session_h.sh head
