#!/bin/bash
mkdir -p gpurun_out/r03l
export PYTHONUNBUFFERED=1
REPO=$(pwd); O=$REPO/gpurun_out/r03l
timeout 1500 python -m pytest tests -m gpu -q --tb=short -p no:cacheprovider > $O/pytest_gpu.log 2>&1; tail -4 $O/pytest_gpu.log | cut -c1-200
timeout 300 python tools/fast_timing.py 8192 4 > $O/fast.log 2>&1; tail -2 $O/fast.log | cut -c1-400
