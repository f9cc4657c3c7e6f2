#!/bin/bash
mkdir -p gpurun_out/r5c18; O=gpurun_out/r5c18; export PYTHONUNBUFFERED=1
timeout 900 python -m pytest tests/test_gpu_training.py tests/test_gpu_loss.py tests/test_gpu_data.py tests/test_gpu_model_golden.py -m gpu -q -x --tb=short -p no:cacheprovider > $O/pytest_train.log 2>&1
echo "pytest exit: $?"; tail -15 $O/pytest_train.log | cut -c1-220
python tools/train_timing.py 256 5 > $O/train_256.log 2>&1; tail -3 $O/train_256.log
python tools/train_timing.py 10 10 > $O/train_10.log 2>&1; tail -3 $O/train_10.log
